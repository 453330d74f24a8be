"""Diagnostic: a sampling profile of ALL host threads over a few training steps (the autograd engine runs the backward functions on its
own thread, which cProfile on the main thread does not see).  Every 0.2 ms: the innermost frame of each thread that lies in this package."""
import collections
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py"] + sys.argv[1:]
import bench  # noqa: E402

a = bench.parse()
if a.mode == "refdefault":
    a.hidden, a.passages, a.passage_len, a.query_len, a.answer_len, a.enc_layers, a.batch, a.mode = 256, 10, 100, 60, 40, 3, 16, "train"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
trainer, opt, sched, batch = bench.build(a, dev)
for _ in range(3):
    trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
inner, leaf, nsamp, stop = collections.Counter(), collections.Counter(), [0], [False]
me = threading.get_ident()


def sampler():
    while not stop[0]:
        for tid, fr in sys._current_frames().items():
            if tid == threading.get_ident():
                continue
            f, first = fr, True
            while f is not None:
                fn = f.f_code.co_filename
                if first:
                    leaf[(tid == me, fn.split("/")[-1], f.f_code.co_name)] += 1
                    first = False
                if "case_rg_amd" in fn:
                    inner[(tid == me, fn.split("case_rg_amd/")[-1], f.f_code.co_name)] += 1
                    break
                f = f.f_back
        nsamp[0] += 1
        time.sleep(0.0002)


th = threading.Thread(target=sampler, daemon=True)
th.start()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop[0] = True
th.join()
print("wall %.2f ms per step, %d samples" % (dt / N * 1e3, nsamp[0]))
per = dt / N * 1e3 / max(1, nsamp[0]) * 1.0
for name, table in (("innermost package frame", inner), ("leaf frame", leaf)):
    print("--", name, "(ms per step, main thread / other threads)")
    for (main, fn, func), c in table.most_common(28):
        print("  %6.2f  %-5s %s:%s" % (c * dt / nsamp[0] / N * 1e3, "main" if main else "bwd", fn, func))
