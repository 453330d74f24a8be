"""Split-KV sweep of the long-memory cross-attention forward (cfg 5: 4 x 40 queries over 20 480 keys, head_dim 96; cfg 2 decoder: 32 x 40 over 3840, head_dim 64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import ops

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (N, h, T, S, d) in ((4, 8, 40, 20480, 96), (32, 8, 40, 3840, 64), (4, 8, 40, 20480, 64)):
    E = h * d
    q = (torch.randn(N, T, E, device="cuda") * 0.5).to(torch.bfloat16)
    kv = (torch.randn(N, S, 2 * E, device="cuda") * 0.5).to(torch.bfloat16)
    valid = torch.ones(N, S, dtype=torch.bool, device="cuda")
    raw = ops._kv_splits
    row = []
    with torch.no_grad():
        for ks in (0, 1, 4, 8, 12, 16, 20, 32, 40, 64):
            ops._kv_splits = raw if ks == 0 else (lambda *a, ks=ks: ks)
            ms = timeit(lambda: ops.attention(q, kv, kv, 0, 0, E, h, d, key_valid=valid))
            row.append("%s:%.4f" % ("policy(%d)" % raw(N, h, T, S, False) if ks == 0 else ks, ms))
    ops._kv_splits = raw
    mb = N * S * 2 * E * 2 / 1e6
    print("N %d S %d d %d (%.0f MB of K/V)  ms by ksplit: %s" % (N, S, d, mb, "  ".join(row)), flush=True)
