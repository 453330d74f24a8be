"""Diagnostic: where the HOST time of one training step goes (cProfile over a few steps of a small geometry, sorted by own time)."""
import cProfile
import os
import pstats
import sys

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
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
