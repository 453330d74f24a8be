"""Split-K sweep of the weight-gradient GEMM (TN, f32 atomics) on the 128x128 and the 256x256 tiling."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import ops
from case_rg_amd import _abi as A

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dt = torch.bfloat16
Mtok = 122880
for (N, K) in [(512, 512), (1536, 512), (512, 2560), (2560, 2560)]:
    g = torch.randn(Mtok, N, device="cuda").to(dt)
    x = torch.randn(Mtok, K, device="cuda").to(dt)
    dw = torch.zeros(N, K, device="cuda", dtype=torch.float32)
    cur = ops._split_for(N, K, Mtok, 2)
    for tile in (128, 256):
        ops.GEMM_TILE = tile
        row = []
        for split in (4, 8, 12, 16, 24, 32, 48, 64):
            ms = timeit(lambda: ops.gemm(g, x, dw, N, K, Mtok, N, K, K, a_kmajor=True, b_kmajor=True, split_k=split, epilogue=A.EPI_ATOMIC))
            row.append("%d:%.3f" % (split, ms))
        print("dW %dx%d tile %d (heuristic split %d)  ms by split: %s" % (N, K, tile, cur, "  ".join(row)), flush=True)
    ops.GEMM_TILE = 0

# atomics against per-split slabs + ordered reduce (case_gemm_dw_slabs) on the 256 tiling, with the fused bias gradient as the step runs it
print("--- 256 tiling, split chosen by ops._split_for: f32 atomics vs slabs ---", flush=True)
for (N, K) in [(512, 512), (1536, 512), (1024, 512), (512, 2560), (2560, 512), (2560, 2560)]:
    g = torch.randn(Mtok, N, device="cuda").to(dt)
    x = torch.randn(Mtok, K, device="cuda").to(dt)
    dw = torch.zeros(N, K, device="cuda", dtype=torch.float32)
    db = torch.zeros(N, device="cuda", dtype=torch.float32)
    split = ops._split_for(N, K, Mtok, 2)
    row = []
    for min_split in (0, 2):
        ops.DW_SLAB_MIN_SPLIT = min_split
        ms = timeit(lambda: ops.gemm(g, x, dw, N, K, Mtok, N, K, K, a_kmajor=True, b_kmajor=True, split_k=split, epilogue=A.EPI_ATOMIC, rowsum_out=db), 20)
        row.append("%s %.3f ms (%.0f TFLOP/s)" % ("slabs" if min_split else "atomics", ms, 2.0 * N * K * Mtok / ms / 1e9))
    print("dW %dx%d split %d: %s" % (N, K, split, "   ".join(row)), flush=True)
    for s2 in (8, 16, 32, 48, 64):
        ms = timeit(lambda: ops.gemm(g, x, dw, N, K, Mtok, N, K, K, a_kmajor=True, b_kmajor=True, split_k=s2, epilogue=A.EPI_ATOMIC, rowsum_out=db), 20)
        print("      slabs split %d: %.3f ms" % (s2, ms), flush=True)
