"""Debug aid: run the 256-tile GEMM on small shapes repeatedly and report where it disagrees with torch (16x16 block map)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import ops

def run(M, N, K, layout, reps=5):
    dt = torch.bfloat16
    torch.manual_seed(1)
    if layout == "nt":
        a, b = torch.randn(M, K, device="cuda").to(dt), (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        ref = a.float() @ b.float().t()
        fn = lambda c: ops.gemm(a, b, c, M, N, K, K, K, N)
    elif layout == "nn":
        a, b = torch.randn(M, K, device="cuda").to(dt), (torch.randn(K, N, device="cuda") * K ** -0.5).to(dt)
        ref = a.float() @ b.float()
        fn = lambda c: ops.gemm(a, b, c, M, N, K, K, N, N, b_kmajor=True)
    else:
        a, b = torch.randn(K, M, device="cuda").to(dt), (torch.randn(K, N, device="cuda") * K ** -0.5).to(dt)
        ref = a.float().t() @ b.float()
        fn = lambda c: ops.gemm(a, b, c, M, N, K, M, N, N, a_kmajor=True, b_kmajor=True)
    ops.GEMM_TILE = 256
    bad_total = 0
    for r in range(reps):
        c = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
        fn(c)
        torch.cuda.synchronize()
        err = (c.float() - ref).abs()
        bad = ~(err <= 0.03 * ref.abs().max())
        if bad.any():
            bad_total += 1
            blocks = bad.reshape(M // 16, 16, N // 16, 16).any(dim=3).any(dim=1)
            idx = blocks.nonzero()
            print("  rep %d: %d bad elements, nan %d, bad 16x16 blocks (row-block, col-block): %s" % (
                r, int(bad.sum()), int(torch.isnan(c.float()).sum()), idx[:24].tolist()))
    print("%s M=%d N=%d K=%d: %d / %d runs bad" % (layout, M, N, K, bad_total, reps))
    ops.GEMM_TILE = 0

for layout in ("nt", "nn", "tn"):
    for (M, N, K) in ((256, 256, 64), (512, 768, 128), (256, 512, 192), (768, 256, 320), (5120, 4096, 192)):
        run(M, N, K, layout)
