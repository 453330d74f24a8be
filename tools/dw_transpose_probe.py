"""Probe: the weight-gradient GEMM dW[N, K] = g^T x with x pre-transposed (A k-major, B k-contiguous) against today's form (both operands
k-major), at the shapes of the 5H blocks of BASELINE cfg 2.  Prints ms per launch for the GEMM alone and for the transposition."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from case_rg_amd import _abi as A  # noqa: E402
from case_rg_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
REP = 6


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


for (M, N, K) in ((122880, 7680, 2560), (122880, 2560, 2560), (122880, 512, 2560), (122880, 2560, 512), (122880, 2048, 512), (122880, 512, 512), (122880, 1536, 512)):
    g = torch.randn(M, N, device=dev).to(torch.bfloat16)
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    split = ops._split_for(N, K, M, 2)
    dw0 = torch.zeros(N, K, device=dev)
    dw1 = torch.zeros(N, K, device=dev)
    b0 = torch.zeros(N, device=dev)
    b1 = torch.zeros(N, device=dev)
    xt = torch.empty(K, M, device=dev, dtype=torch.bfloat16)

    def tn():
        ops.gemm(g, x, dw0, N, K, M, N, K, K, a_kmajor=True, b_kmajor=True, split_k=split, epilogue=A.EPI_ATOMIC, rowsum_out=b0)

    def tr():
        xt.copy_(x.t())

    def tn_t():
        ops.gemm(g, xt, dw1, N, K, M, N, M, K, a_kmajor=True, b_kmajor=False, split_k=split, epilogue=A.EPI_ATOMIC, rowsum_out=b1)

    t_tn = timed(tn)
    t_tr = timed(tr)
    t_nt = timed(tn_t)
    dw0.zero_(); dw1.zero_(); b0.zero_(); b1.zero_()
    tn(); tr(); tn_t()
    torch.cuda.synchronize()
    err = ((dw0 - dw1).norm() / dw0.norm()).item()
    fl = 2.0 * M * N * K
    print("out %5d x in %5d over %d tokens split %2d:  both k-major %.3f ms (%.0f TFLOP/s)   x^T given %.3f ms (%.0f)   torch transpose %.3f ms   rel diff %.1e"
          % (N, K, M, split, t_tn, fl / t_tn / 1e9, t_nt, fl / t_nt / 1e9, t_tr, err), flush=True)
    del g, x, xt, dw0, dw1
