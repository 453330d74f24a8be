"""Diagnostic: which lines of the package still allocate zero-filled tensors (one fill launch each) during a training step."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1"]
import bench  # noqa: E402

a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
trainer, opt, sched, batch = bench.build(a, dev)
for _ in range(3):
    trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
cnt = collections.Counter()


def wrap(mod, name):
    raw = getattr(mod, name)

    def f(*args, **kw):
        st = [fr for fr in traceback.extract_stack()[:-1] if "case_rg_amd" in fr.filename or "bench.py" in fr.filename]
        where = "%s:%d" % (st[-1].filename.split("/case_rg_amd/")[-1], st[-1].lineno) if st else "(elsewhere)"
        cnt[(name, where)] += 1
        return raw(*args, **kw)

    setattr(mod, name, f)


for n in ("zeros", "zeros_like"):
    wrap(torch, n)
wrap(torch.Tensor, "zero_")
wrap(torch.Tensor, "new_zeros")
trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print("%4d %-12s %s" % (c, name, where))
