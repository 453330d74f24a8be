"""Round 6: where does the fused flash-style BACKWARD at head_dim 320 / 480 lose accuracy?
Relative L2 error of dQ, dK, dV (separately) against f32 autograd on the same bf16 inputs, fused backward against the GEMM -> softmax ->
GEMM backward, at several input scales (the error of an attention backward in bf16 grows with the sharpness of the softmax).
    python tools/wide_bwd_bisect.py > gpurun_out/wide_bwd_bisect.txt"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi, ops  # noqa: E402

DEV = "cuda"


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def case(N, h, L, d, scale, fused, lens=None):
    E = h * d
    g0 = torch.Generator().manual_seed(1)
    q = (torch.randn(N, L, E, generator=g0) * scale).to(DEV).to(torch.bfloat16).requires_grad_()
    k = (torch.randn(N, L, E, generator=g0) * scale).to(DEV).to(torch.bfloat16).requires_grad_()
    v = (torch.randn(N, L, E, generator=g0) * scale).to(DEV).to(torch.bfloat16).requires_grad_()
    g = torch.randn(N, L, E, generator=g0).to(DEV).to(torch.bfloat16)
    saved = _abi.lib.case_attention_supported
    ops.ATTENTION_MODE = "fused"
    try:
        if not fused:
            _abi.lib.case_attention_supported = lambda _d: 0
        valid = None
        if lens is not None:
            valid = torch.arange(L, device=DEV)[None, :] < torch.tensor(lens, device=DEV)[:, None]
        o = ops.attention(q, k, v, 0, 0, 0, h, d, key_valid=valid)
        o.backward(g)
    finally:
        _abi.lib.case_attention_supported = saved
        ops.ATTENTION_MODE = "auto"
    rq, rk, rv = [t.detach().float().requires_grad_() for t in (q, k, v)]
    qh, kh, vh = [t.reshape(N, L, h, d).transpose(1, 2) for t in (rq, rk, rv)]
    sc = qh @ kh.transpose(-1, -2) / math.sqrt(d)
    if valid is not None:
        sc = sc.masked_fill(~valid[:, None, None, :], float('-inf'))
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(N, L, E)
    ref.backward(g.float())
    return rel(o, ref), rel(q.grad, rq.grad), rel(k.grad, rk.grad), rel(v.grad, rv.grad)


print("%-5s %-6s %-8s %9s %9s %9s %9s" % ("d", "scale", "path", "o", "dq", "dk", "dv"))
for d, h in ((64, 8), (320, 8), (480, 4)):
    for scale in (0.3, 0.7, 1.5):
        for fused in (True, False):
            e = case(2, h, 384, d, scale, fused)
            print("%-5d %-6.1f %-8s %9.2e %9.2e %9.2e %9.2e" % (d, scale, "fused" if fused else "unfused", *e), flush=True)

print("ragged key masks: valid lengths 384 / 200 / 2 (a filler passage), 3 sequences")
for d, h in ((64, 8), (320, 8), (480, 4)):
    for fused in (True, False):
        e = case(3, h, 384, d, 0.7, fused, lens=[384, 200, 2])
        print("%-5d %-6.1f %-8s %9.2e %9.2e %9.2e %9.2e" % (d, 0.7, "fused" if fused else "unfused", *e), flush=True)
