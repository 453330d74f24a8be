"""Times the fused encoder chain kernels (csrc/encoder_chain.hip) alone on random data: python tools/chain_bench.py [rows] [iters]
Prints one JSON line per variant: ms, TFLOP/s of the GEMM work it contains, fraction of the bf16 MFMA peak.
CASE_HIP_LIB selects another build of the library (A/B of kernel variants)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import case_rg_amd  # noqa: E402
from case_rg_amd import ops  # noqa: E402
from case_rg_amd.utils import fill_params  # noqa: E402

PEAK = 2516.6


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 10 * 384
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    L = 384
    N = rows // L
    dev = torch.device("cuda", 0)
    case_rg_amd.set_compute_dtype(torch.bfloat16)
    ns = case_rg_amd.namespace()
    layer = ns.TransformerEncoderLayer(512, 8, dim_feedforward=512, dropout=0.1, activation="gelu")
    enc = fill_params(ns.TransformerEncoder(layer, 2), 3, gain=2.0).to(dev).eval()
    x = torch.randn(N, L, 512, device=dev).to(torch.bfloat16)
    s = torch.randn(N, L, 512, device=dev).to(torch.bfloat16)
    l0, l1 = enc.layers[0], enc.layers[1]
    work = {"head": 2 * 512 * 1536, "full": 2 * 512 * (3 * 512 + 1536), "tail": 2 * 512 * 3 * 512}
    with torch.no_grad():
        for variant, args in (("head", (x, None, None, l0)), ("full", (x, s, l0, l1)), ("tail", (x, s, l1, None))):
            for _ in range(3):
                ops.encoder_chain(variant, *args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.encoder_chain(variant, *args)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            tf = work[variant] * N * L / ms / 1e9
            print(json.dumps({"variant": variant, "rows": N * L, "ms": round(ms, 4), "tflops": round(tf, 1), "frac": round(tf / PEAK, 4),
                              "lib": os.environ.get("CASE_HIP_LIB", "default")}))


if __name__ == "__main__":
    main()
