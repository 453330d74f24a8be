"""Diagnostic: shapes and (forward) call sites of the elementwise adds / fills / copies one training step launches."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1"]
import bench  # noqa: E402

a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
trainer, opt, sched, batch = bench.build(a, dev)
for _ in range(2):
    trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    trainer.train_batch(0, dict(batch), "train", opt, sched)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::zero_", "aten::cat", "aten::mul", "aten::sum"):
        st = [s for s in (e.stack or []) if "case_rg_amd" in s or "bench.py" in s]
        where = st[0].split("/case_rg_amd/")[-1][:70] if st else "(autograd engine)"
        cnt[(e.name, str(e.input_shapes)[:90], where)] += 1
for (name, shp, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print("%4d %-16s %-92s %s" % (c, name, shp, where))
