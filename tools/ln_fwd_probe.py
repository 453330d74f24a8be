"""Probe: LayerNorm forward (inference form) at the greedy pass's and the training step's row counts: ms and TB/s (read + write)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from case_rg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, rep=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep


with torch.no_grad():
    for rows, C in ((983040, 2560), (983040, 512), (122880, 2560), (122880, 512)):
        x = torch.randn(rows, C, device=dev).to(torch.bfloat16)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        t = timed(lambda: ops.layer_norm(x, g, b, 1e-5))
        y = torch.empty_like(x)
        tc = timed(lambda: y.copy_(x))
        print("rows %7d x %4d: LayerNorm %.3f ms = %.2f TB/s   device copy %.3f ms = %.2f TB/s" % (rows, C, t, 2 * x.numel() * 2 / t / 1e9, tc, 2 * x.numel() * 2 / tc / 1e9), flush=True)
        del x, y
