"""Per-phase cycles of K17 (csrc/attn_scores.hip built with -DSC_STAMPS [-DSC_STAMP_WAVE=w], CASE_HIP_LIB=...): s_memtime stamps of one
wave on each workgroup's second item, median over workgroups.  argv[1]: fwd | bwd."""
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi as A, ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
N, h, L, d = 320, 8, 384, 320
E = h * d
qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.5).to(torch.bfloat16)
dO = torch.randn(N, L, E, device="cuda").to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.uint8, device="cuda")
drop = (0.1, 11, 0)
ad = ops._attn_desc(N, h, L, L, d, qkv, qkv, qkv, False, 1.0 / math.sqrt(d), drop)
P, Pd, dS = (torch.empty(N, h, L, L, dtype=torch.bfloat16, device="cuda") for _ in range(3))
for _ in range(3):
    A.call("case_attention_scores_fwd", ad, ops._ptr(qkv, 0), ops._ptr(qkv, E), ops._ptr(valid), ops._ptr(P), ops._ptr(Pd), ops._stream())
    if which == "bwd":
        A.call("case_attention_scores_bwd", ad, ops._ptr(dO), ops._ptr(qkv, 2 * E), ops._ptr(P), ops._ptr(dS), ops._stream())
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * (256 * 16))()
assert A.lib.case_attention_scores_stamps(buf) == 0
raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(256, 16)
names = {0: "item start", 1: "K step 0 ready", 2: "K step 1 ready", 3: "K step 2 ready", 4: "K step 3 ready", 5: "K step 4 ready",
         8: "K loop done", 10: "statistics exchanged", 11: "epilogue done"}
ids = [i for i in sorted(names) if np.median(raw[:, i]) > 0]
tot = 0
for a, b in zip(ids[:-1], ids[1:]):
    dt = np.median(raw[:, b] - raw[:, a])
    tot += dt
    print("%-26s %8.0f cycles" % (names[b], dt))
print("sum %.0f" % tot)
