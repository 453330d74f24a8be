"""Soak of K17 in ONE process: 40 rounds of the cfg-2-geometry forward / backward / products on fresh random inputs, each round checked
bit for bit against a second launch of the same inputs (no atomics anywhere: any difference is a race)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi as A, ops  # noqa: E402

N, h, L, d = 96, 8, 384, 320
E = h * d
bad = 0
for it in range(40):
    g = torch.Generator(device="cuda").manual_seed(1000 + it)
    qkv = (torch.randn(N, L, 3 * E, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    dO = torch.randn(N, L, E, device="cuda", generator=g).to(torch.bfloat16)
    valid = (torch.rand(N, L, device="cuda", generator=g) > 0.2).to(torch.uint8)
    drop = (0.1, 7 + it, 2 * it)
    ad = ops._attn_desc(N, h, L, L, d, qkv, qkv, qkv, False, 1.0 / math.sqrt(d), drop)
    outs = []
    for rep in range(2):
        P, Pd, dS = (torch.empty(N, h, L, L, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        O = torch.empty(N, L, E, dtype=torch.bfloat16, device="cuda")
        G = torch.empty(N, L, 3 * E, dtype=torch.bfloat16, device="cuda")
        A.call("case_attention_scores_fwd", ad, ops._ptr(qkv, 0), ops._ptr(qkv, E), ops._ptr(valid), ops._ptr(P), ops._ptr(Pd), ops._stream())
        A.call("case_attention_scores_bwd", ad, ops._ptr(dO), ops._ptr(qkv, 2 * E), ops._ptr(P), ops._ptr(dS), ops._stream())
        assert ops.AttentionFn._product(Pd, qkv, 2 * E, O, 0, h, d, L, L, False)
        assert ops.AttentionFn._product(dS, qkv, 0, G, E, h, d, L, L, True, 0.5)
        outs.append((P, Pd, dS, O, G[:, :, E:2 * E].clone()))
    same = all(torch.equal(a, b) for a, b in zip(*outs))
    finite = all(torch.isfinite(t.float()).all().item() for t in outs[0])
    bad += (not same) or (not finite)
    if it % 10 == 9:
        print("round", it + 1, "mismatching rounds so far:", bad, flush=True)
print("K17 soak:", "OK" if bad == 0 else "%d BAD ROUNDS" % bad)
sys.exit(1 if bad else 0)
