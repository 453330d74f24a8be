"""Calibration only (never on the product path): what the vendor GEMM (hipBLASLt behind torch.matmul) reaches on this box at the
shapes of CaSE cfg 2, next to this library's kernels on the same tensors.  Tells how much of the gap to the 2.5 PFLOP/s bf16 peak
is the chip's clock under MFMA load and how much is ours.
    python tools/blas_ceiling.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev, dt = "cuda", torch.bfloat16
    out = []
    for (M, K, N) in [(122880, 2560, 7680), (122880, 2560, 2560), (122880, 2560, 512), (122880, 512, 1536), (122880, 512, 512),
                      (245760, 512, 1536), (245760, 512, 512), (8192, 8192, 8192)]:
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        fl = 2.0 * M * N * K
        t_ours = timeit(lambda: ops.gemm(x, w, y, M, N, K, K, K, N))
        wt = w.t()
        t_blas = timeit(lambda: torch.matmul(x, wt, out=y))
        rec = {"M": M, "K": K, "N": N, "case_gemm_tflops": round(fl / t_ours / 1e12, 1), "hipblaslt_tflops": round(fl / t_blas / 1e12, 1)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del x, w, y


if __name__ == "__main__":
    main()
