"""Probe: is the K = 512 GEMM bound per CU or by the chip's HBM?  The same 122 880 x N x 512 launch on 256 / 192 / 128 / 64 CUs
(case_set_reserved_cus shrinks the persistent grid): per-CU-bound work scales with 1 / CUs, HBM-bound work does not."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from case_rg_amd import _abi as A  # noqa: E402
from case_rg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
dt = torch.bfloat16
M, K = 122880, 512


def timed(fn, rep=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


for N in (512, 2048):
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(dt)
    for reserve in (0, 64, 128):
        A.call("case_set_reserved_cus", reserve)
        t0 = timed(lambda: ops.gemm(x, w, y, M, N, K, K, K, N))
        t1 = timed(lambda: ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL, bias_col=bias, aux=res, ld_aux=N, drop=(0.1, 1, 0)))
        tiles = (M // 256) * (N // 256)
        cus = 256 - reserve
        print("N %4d on %3d CUs (%.2f rounds): plain %6.1f us = %5.1f us per round   bias+res+drop %6.1f us = %5.1f us per round"
              % (N, cus, tiles / cus, t0, t0 / -(-tiles // cus), t1, t1 / -(-tiles // cus)), flush=True)
    A.call("case_set_reserved_cus", 0)
    del x, w, y, res
