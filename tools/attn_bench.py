#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernels on the CaSE cfg 2 shapes (320 sequences x 8 heads x 384 tokens)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import config, ops  # noqa: E402


def timeit(fn, iters=5):
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
    p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0   # python tools/attn_bench.py 0.1 -> with attention dropout
    config.set_dropout(p_drop > 0.0)
    for d in (320, 64):
        N, h, L = 320, 8, 384
        E = h * d
        qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_()
        valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
        g = torch.randn(N, L, E, device="cuda").to(torch.bfloat16)
        fl = 4.0 * N * h * L * L * d
        t = timeit(lambda: ops.attention(qkv.detach(), qkv.detach(), qkv.detach(), 0, E, 2 * E, h, d, key_valid=valid, p_drop=p_drop))
        print("fwd d=%3d  %7.3f ms  %7.1f TFLOP/s" % (d, t * 1e3, fl / t / 1e12))

        def fb():
            o = ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, p_drop=p_drop)
            o.backward(g)
        t2 = timeit(fb)
        print("fwd+bwd d=%3d  %7.3f ms  (bwd %7.3f ms, %7.1f TFLOP/s on 2.5x fwd flops)" % (d, t2 * 1e3, (t2 - t) * 1e3, 2.5 * fl / (t2 - t) / 1e12))


if __name__ == "__main__":
    main()
