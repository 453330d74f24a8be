"""Micro-benchmark of the fused attention kernels: self-attention on the cfg 2 / cfg 5 passage shapes and the long-memory
cross-attention of the decoder.  `python tools/attn_bench.py [p_drop]`; one JSON line per case."""
import json
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


def self_attention(N, h, L, d, p_drop):
    E = h * d
    qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_()
    valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
    g = torch.randn(N, L, E, device="cuda").to(torch.bfloat16)
    fl = 4.0 * N * h * L * L * d
    t = timeit(lambda: ops.attention(qkv.detach(), qkv.detach(), qkv.detach(), 0, E, 2 * E, h, d, key_valid=valid, p_drop=p_drop))

    def fb():
        ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, p_drop=p_drop).backward(g)
    t2 = timeit(fb)
    print(json.dumps({"case": "self", "N": N, "heads": h, "L": L, "head_dim": d, "p_drop": p_drop, "mode": ops.ATTENTION_MODE,
                      "fwd_ms": round(t * 1e3, 3), "fwd_tflops": round(fl / t / 1e12, 1), "bwd_ms": round((t2 - t) * 1e3, 3),
                      "bwd_tflops_on_2.5x_fwd": round(2.5 * fl / (t2 - t) / 1e12, 1)}))


def cross_attention(N, h, Lq, S, d):
    """decoder cross-attention over a long memory: HBM-bound on the K/V stream (2 S H 2 bytes per sequence)."""
    E = h * d
    q = (torch.randn(N, Lq, E, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_()
    kv = (torch.randn(N, S, 2 * E, device="cuda") * 0.5).to(torch.bfloat16).requires_grad_()
    valid = torch.ones(N, S, dtype=torch.bool, device="cuda")
    g = torch.randn(N, Lq, E, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: ops.attention(q.detach(), kv.detach(), kv.detach(), 0, 0, E, h, d, key_valid=valid), iters=10)

    def fb():
        ops.attention(q, kv, kv, 0, 0, E, h, d, key_valid=valid).backward(g)
    t2 = timeit(fb, iters=10)
    stream = N * S * 2 * E * 2
    print(json.dumps({"case": "cross", "N": N, "heads": h, "Lq": Lq, "S": S, "head_dim": d, "mode": ops.ATTENTION_MODE,
                      "fwd_ms": round(t * 1e3, 4), "kv_stream_MB": round(stream / 1e6, 1), "fwd_GBps": round(stream / t / 1e9, 1),
                      "fwd_frac_of_8TBps": round(stream / t / 8e12, 4), "bwd_ms": round((t2 - t) * 1e3, 4)}))


def main():
    p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    config.set_dropout(p_drop > 0.0)
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["auto"]
    for mode in modes:
        ops.ATTENTION_MODE = mode
        for (N, h, L, d) in ((320, 8, 384, 320), (320, 8, 384, 64), (160, 8, 512, 480), (160, 8, 512, 96)):
            self_attention(N, h, L, d, p_drop)
        if p_drop == 0.0:
            cross_attention(32, 8, 40, 3840, 64)     # cfg 2 decoder, passage memory
            cross_attention(4, 8, 40, 20480, 96)     # cfg 5: 40 x 512 memory, d_model 768, 4 sequences per GPU
            cross_attention(256, 8, 1, 3840, 64)     # cfg 4 greedy step


if __name__ == "__main__":
    main()
