import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1"]
import bench
a = bench.parse()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
trainer, opt, sched, batch = bench.build(a, dev)
for _ in range(2): trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    trainer.train_batch(0, dict(batch), "train", opt, sched)
    torch.cuda.synchronize()
import collections
cnt = collections.Counter(); stacks = collections.defaultdict(collections.Counter)
for e in prof.events():
    if e.name in ("aten::copy_", "aten::zeros", "aten::zero_", "aten::fill_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::cat", "aten::sum", "aten::mul", "aten::to", "aten::_to_copy", "aten::zeros_like", "aten::empty"):
        cnt[e.name] += 1
        st = [s for s in (e.stack or []) if "case_rg_amd" in s or "bench.py" in s]
        stacks[e.name][st[0].split("/root/repo/")[-1] if st else "?"] += 1
for n, c in cnt.most_common():
    print(n, c)
    for s, k in stacks[n].most_common(8): print("     ", k, s[:110])
