"""Diagnostic: which forward tensors have several consumers in one CaSE training step -- the autograd engine sums their gradients with
one elementwise add per extra consumer (the `aten::add_` launches of the step profile).  Walks the graph from the loss and prints, for
every (producer node, output index) referenced more than once, the producer, its consumers and the gradient shape."""
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
trainer.train_batch(0, dict(batch), "train", opt, sched)
loss = trainer.model(dict(batch), method="train")
parts = torch.cat([l.mean().reshape(1) for l in loss]) if isinstance(loss, (tuple, list)) else loss.mean().reshape(1)
root = parts.sum().grad_fn

users = collections.defaultdict(list)
seen, stack = {root}, [root]
while stack:
    fn = stack.pop()
    for nxt, idx in fn.next_functions:
        if nxt is None:
            continue
        users[(nxt, idx)].append(type(fn).__name__)
        if nxt not in seen:
            seen.add(nxt)
            stack.append(nxt)


def shape_of(fn, idx):
    try:
        m = fn._input_metadata[idx]
        return tuple(m.shape), str(m.dtype).replace("torch.", "")
    except Exception:
        return None, None


rows = collections.Counter()
for (fn, idx), us in users.items():
    if len(us) > 1 and type(fn).__name__ != "AccumulateGrad":
        shp, dt = shape_of(fn, idx)
        rows[(type(fn).__name__, idx, str(shp), dt, " + ".join(sorted(us)))] += 1
print("%d graph nodes; tensors with several consumers (count, producer[output], gradient shape, consumers):" % len(seen))
for (name, idx, shp, dt, us), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    print("%3d  %-34s[%d] %-22s %-9s <- %s" % (c, name, idx, shp, dt, us))
acc = collections.Counter()
for (fn, idx), us in users.items():
    if len(us) > 1 and type(fn).__name__ == "AccumulateGrad":
        acc[(tuple(fn.variable.shape), len(us))] += 1
print("parameters used more than once (shape, uses) x count:", dict(acc))
