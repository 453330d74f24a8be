"""Micro-benchmark of the MFMA GEMM family on the shapes that dominate CaSE cfg 2 (random bf16 data).
    python tools/gemm_bench.py            # prints TFLOP/s per shape / layout
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import ops  # noqa: E402
from case_rg_amd import _abi as A  # noqa: E402


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
    dev = "cuda"
    dt = torch.bfloat16
    M = 122880
    rows = []
    for (K, N) in [(2560, 7680), (2560, 2560), (2560, 512), (512, 1536), (512, 512)]:
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        g = torch.randn(M, N, device=dev).to(dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        dx = torch.empty(M, K, device=dev, dtype=dt)
        dw = torch.zeros(N, K, device=dev, dtype=torch.float32)
        fl = 2.0 * M * N * K
        t = timeit(lambda: ops.gemm(x, w, y, M, N, K, K, K, N))
        rows.append(("NT  fwd  M=%d K=%d N=%d" % (M, K, N), fl / t / 1e12, t * 1e3))
        t = timeit(lambda: ops.gemm(g, w, dx, M, K, N, N, K, K, b_kmajor=True))
        rows.append(("NN  dX   M=%d K=%d N=%d" % (M, N, K), fl / t / 1e12, t * 1e3))
        split = ops._split_for(N, K, M, 2)
        t = timeit(lambda: ops.gemm(g, x, dw, N, K, M, N, K, K, a_kmajor=True, b_kmajor=True, split_k=split, epilogue=A.EPI_ATOMIC))
        rows.append(("TN  dW   M=%d K=%d N=%d split=%d" % (N, M, K, split), fl / t / 1e12, t * 1e3))
        if K == 512:  # the epilogues these shapes carry in the model (bias + residual + dropout; GELU derivative + dropout)
            bias = torch.randn(N, device=dev)
            res = torch.randn(M, N, device=dev).to(dt)
            z = torch.randn(M, K, device=dev).to(dt)
            E = A.EPI_BIAS_COL | A.EPI_RESIDUAL
            t = timeit(lambda: ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=E, bias_col=bias, aux=res, ld_aux=N, drop=(0.1, 1, 0)))
            rows.append(("NT  fwd + bias + residual + dropout  K=%d N=%d" % (K, N), fl / t / 1e12, t * 1e3))
            t = timeit(lambda: ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_GELU, bias_col=bias, aux_out=res, ld_aux=N, drop=(0.1, 1, 0)))
            rows.append(("NT  fwd + bias + GELU(+z out) + dropout K=%d N=%d" % (K, N), fl / t / 1e12, t * 1e3))
            t = timeit(lambda: ops.gemm(g, w, dx, M, K, N, N, K, K, b_kmajor=True, epilogue=A.EPI_MUL_DGELU, aux=z, ld_aux=K, drop=(0.1, 1, 0)))
            rows.append(("NN  dX * gelu'(z) + dropout          N=%d K=%d" % (N, K), fl / t / 1e12, t * 1e3))
            t = timeit(lambda: ops.gemm(g, w, dx, M, K, N, N, K, K, b_kmajor=True, epilogue=A.EPI_RESIDUAL, aux=z, ld_aux=K))
            rows.append(("NN  dX + residual                    N=%d K=%d" % (N, K), fl / t / 1e12, t * 1e3))
            del bias, res, z
        del x, w, g, y, dx, dw
    # attention-shaped batched products: 320 sequences x 8 heads, L = 384
    for d in (320, 64):
        Nb, h, L = 320, 8, 384
        E = h * d
        qkv = torch.randn(Nb, L, 3 * E, device=dev).to(dt)
        S = torch.empty(Nb, h, L, L, device=dev, dtype=dt)
        O = torch.empty(Nb, L, E, device=dev, dtype=dt)
        fl = 2.0 * Nb * h * L * L * d
        t = timeit(lambda: ops.gemm(qkv, qkv, S, L, L, d, 3 * E, 3 * E, L, a_off=0, b_off=E, batch1=Nb, batch2=h,
                                    sa=(L * 3 * E, d), sb=(L * 3 * E, d), sc=(h * L * L, L * L), alpha=d ** -0.5))
        rows.append(("QK^T d=%d" % d, fl / t / 1e12, t * 1e3))
        t = timeit(lambda: ops.gemm(S, qkv, O, L, d, L, L, 3 * E, E, b_off=2 * E, b_kmajor=True, batch1=Nb, batch2=h,
                                    sa=(h * L * L, L * L), sb=(L * 3 * E, d), sc=(L * E, d)))
        rows.append(("PV   d=%d" % d, fl / t / 1e12, t * 1e3))
    for name, tf, ms in rows:
        print("%-48s %8.1f TFLOP/s  %8.3f ms" % (name, tf, ms))


if __name__ == "__main__":
    main()
