"""Diagnostic: when does each GradSync bucket launch during backward -- from a hook (overlappable) or from finish()'s flush (exposed)?
Forced one-rank RCCL group at BASELINE cfg 2.  Prints per bucket: size, parameters without a gradient, how it was launched."""
import os
import sys

os.environ["CASE_FORCE_GRADSYNC"] = "1"
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
import bench  # noqa: E402

a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
import torch.distributed as dist  # noqa: E402

dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29578", rank=0, world_size=1, device_id=dev)
trainer, opt, sched, batch = bench.build(a, dev)
sync = trainer.sync
print("sync active:", sync is not None and sync.active, "buckets:", len(sync.buckets) if sync else 0)
log = []
raw_ready, raw_launch = sync._launch_ready, sync._launch
state = {"flush": False}


def ready(flush=False):
    state["flush"] = flush
    return raw_ready(flush)


def launch(b):
    idx = next(i for i, x in enumerate(sync.buckets) if x is b)
    log.append((idx, state["flush"], sum(1 for p, _, _ in b["items"] if p.grad is None)))
    return raw_launch(b)


sync._launch_ready, sync._launch = ready, launch
for _ in range(2):
    del log[:]
    trainer.train_batch(0, dict(batch), "train", opt, sched)
torch.cuda.synchronize()
names = {id(p): n for n, p in trainer.model.named_parameters()}
for idx, flush, missing in log:
    b = sync.buckets[idx]
    none = [names[id(p)] for p, _, _ in b["items"] if id(p) in names and p.grad is None]
    print("bucket %2d  %6.1f MB  %3d tensors  launched %s  tensors without a gradient at launch: %d" % (
        idx, b["flat"].numel() * 4 / 2 ** 20, len(b["items"]), "by finish() (exposed)" if flush else "by a hook (overlaps backward)", missing))
