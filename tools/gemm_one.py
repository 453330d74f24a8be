"""One GEMM shape, a few launches: target for rocprofv3 --pmc runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import ops  # noqa: E402

M, K, N = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (122880, 2560, 7680))]
mode = sys.argv[4] if len(sys.argv) > 4 else "nt"
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    if mode == "nt":
        ops.gemm(x, w, y, M, N, K, K, K, N)
    elif mode == "nt_epi":  # bias + residual + dropout: the out-projection / FFN2 form of the encoder layers
        from case_rg_amd import _abi as A
        if "res" not in globals():
            res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
            bias = torch.randn(N, device="cuda")
        ops.gemm(x, w, y, M, N, K, K, K, N, epilogue=A.EPI_BIAS_COL | A.EPI_RESIDUAL, bias_col=bias, aux=res, ld_aux=N, drop=(0.1, 1, 0))
    elif mode == "nn":
        ops.gemm(y, w, x, M, K, N, N, K, K, b_kmajor=True)
    else:  # tn: dW[N, K] = y^T x
        from case_rg_amd import _abi as A
        dw = torch.zeros(N, K, device="cuda")
        ops.gemm(y, x, dw, N, K, M, N, K, K, a_kmajor=True, b_kmajor=True, split_k=2, epilogue=A.EPI_ATOMIC)
torch.cuda.synchronize()
