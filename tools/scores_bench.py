"""K17 (attention scores with the softmax in the GEMM) against the two-kernel path it replaces, at the cfg 2 block geometry
(320 sequences x 8 heads x 384 x 384, head_dim 320, dropout 0.1): ms per call of the forward (scores -> P, Pd) and backward (dO, V, P -> dS)."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi as A, ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


N, h, L, d = 320, 8, 384, 320
E = h * d
qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.5).to(torch.bfloat16)
dO = torch.randn(N, L, E, device="cuda").to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.uint8, device="cuda")
drop = (0.1, 11, 0)
alpha = 1.0 / math.sqrt(d)
ad = ops._attn_desc(N, h, L, L, d, qkv, qkv, qkv, False, alpha, drop)
P, Pd, dS = (torch.empty(N, h, L, L, dtype=torch.bfloat16, device="cuda") for _ in range(3))
S = torch.empty(N, h, L, L, dtype=torch.float32, device="cuda")
flops = 2.0 * N * h * L * L * d


def fwd_new():
    A.call("case_attention_scores_fwd", ad, ops._ptr(qkv, 0), ops._ptr(qkv, E), ops._ptr(valid), ops._ptr(P), ops._ptr(Pd), ops._stream())


def bwd_new():
    A.call("case_attention_scores_bwd", ad, ops._ptr(dO), ops._ptr(qkv, 2 * E), ops._ptr(P), ops._ptr(dS), ops._stream())


def fwd_old():
    ops.gemm(qkv, qkv, S, L, L, d, 3 * E, 3 * E, L, a_off=0, b_off=E, batch1=N, batch2=h, sa=(L * 3 * E, d), sb=(L * 3 * E, d),
             sc=(h * L * L, L * L), alpha=alpha)
    sd = ops._softmax_desc(N, h, L, L, False, A.F32, A.BF16, drop)
    A.call("case_softmax_fwd", sd, ops._ptr(S), ops._ptr(valid), None, ops._ptr(P), ops._ptr(Pd), ops._stream())


def bwd_old():
    ops.gemm(dO, qkv, dS, L, L, d, E, 3 * E, L, b_off=2 * E, batch1=N, batch2=h, sa=(L * E, d), sb=(L * 3 * E, d), sc=(h * L * L, L * L))
    sd = ops._softmax_desc(N, h, L, L, False, A.BF16, A.BF16, drop)
    A.call("case_softmax_bwd", sd, ops._ptr(dS), ops._ptr(P), ops._ptr(dS), ops._stream())


O = torch.empty(N, L, E, dtype=torch.bfloat16, device="cuda")
G3 = torch.empty(N, L, 3 * E, dtype=torch.bfloat16, device="cuda")
pstr = (h * L * L, L * L)


def pv_new():
    assert ops.AttentionFn._product(Pd, qkv, 2 * E, O, 0, h, d, L, L, False)


def pv_old():
    ops.gemm(Pd, qkv, O, L, d, L, L, 3 * E, E, b_off=2 * E, b_kmajor=True, batch1=N, batch2=h, sa=pstr, sb=(L * 3 * E, d), sc=(L * E, d))


def dv_new():
    assert ops.AttentionFn._product(Pd, dO, 0, G3, 2 * E, h, d, L, L, True)


def dv_old():
    ops.gemm(Pd, dO, G3, L, d, L, L, E, 3 * E, c_off=2 * E, a_kmajor=True, b_kmajor=True, batch1=N, batch2=h, sa=pstr, sb=(L * E, d),
             sc=(L * 3 * E, d))


for name, fn in (("pv_gemm128", pv_old), ("pv_k17", pv_new), ("dv_gemm128", dv_old), ("dv_k17", dv_new), ("fwd_two_kernels", fwd_old), ("fwd_k17", fwd_new), ("bwd_two_kernels", bwd_old), ("bwd_k17", bwd_new)):
    ms = timeit(fn)
    print(json.dumps({"case": name, "N": N, "heads": h, "L": L, "head_dim": d, "p_drop": 0.1, "ms": round(ms, 4),
                      "gemm_tflops": round(flops / ms / 1e9, 1)}))
