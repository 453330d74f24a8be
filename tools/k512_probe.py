"""Probe: the K = 512 GEMMs of BASELINE cfg 2 (122 880 rows) under the 256 x 256 persistent tiling and the 128 x 128 two-workgroups-per-CU
tiling, plain and with the epilogues the model uses, against their two bounds (MFMA time at the sustained rate, bytes at the measured copy rate)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from case_rg_amd import _abi as A  # noqa: E402
from case_rg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
dt = torch.bfloat16
REP = 20


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e-3


M, K = 122880, 512
a = torch.randn(M, 2048, device=dev).to(dt)
b = torch.empty_like(a)
t = timed(lambda: b.copy_(a))
print("copy of %d MB: %.1f us = %.2f TB/s (read + write)" % (a.numel() * 2 >> 20, t * 1e6, 2 * a.numel() * 2 / t / 1e12), flush=True)
del a, b
for N in (512, 1536, 2048):
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(dt)
    fl = 2.0 * M * N * K
    for tile in (256, 128):
        ops.GEMM_TILE = tile
        for name, fn, nbytes in (
                ("plain", lambda: ops.gemm(x, w, y, M, N, K, K, K, N), (M * K + N * K + M * N) * 2),
                ("bias", lambda: ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL, bias_col=bias), (M * K + N * K + M * N) * 2),
                ("bias+res+drop", lambda: ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL, bias_col=bias, aux=res, ld_aux=N,
                                                   drop=(0.1, 1, 0)), (M * K + N * K + 2 * M * N) * 2)):
            t = timed(fn)
            print("N %4d tile %3d %-14s %7.1f us  %6.0f TFLOP/s  %5.2f TB/s algorithmic" % (N, tile, name, t * 1e6, fl / t / 1e12, nbytes / t / 1e12), flush=True)
    ops.GEMM_TILE = 0
    del x, w, y, res
