"""Fixed vs per-K-tile cost of the GEMM tilings: time M x N x K for growing K (NT, bf16).  python tools/gemm_ksweep.py [N]"""
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
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    M = 122880
    dt = torch.bfloat16
    for policy in (128, 256):
        ops.GEMM_TILE = policy
        for K in (64, 128, 256, 512, 1024, 2048):
            x = torch.randn(M, K, device="cuda").to(dt)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
            y = torch.empty(M, N, device="cuda", dtype=dt)
            t = timeit(lambda: ops.gemm(x, w, y, M, N, K, K, K, N))
            print("tile %d  N=%d K=%5d  %8.3f ms  %7.1f TFLOP/s" % (policy, N, K, t * 1e3, 2.0 * M * N * K / t / 1e12))
    ops.GEMM_TILE = 0


if __name__ == "__main__":
    main()
