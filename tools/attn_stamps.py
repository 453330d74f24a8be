"""Diagnostic: per-phase cycle shares of the slab attention forward (library built with -DFAS_STAMPS)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import _abi, ops  # noqa: E402

N, h, L, d = 320, 8, 384, int(sys.argv[1]) if len(sys.argv) > 1 else 320
E = h * d
qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.5).to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
nblk = N * h * 3
buf = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
_abi.lib.case_debug_stamp_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3):
    ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid)
_abi.lib.case_debug_stamp_buffer(buf.data_ptr())
ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid)
torch.cuda.synchronize()
_abi.lib.case_debug_stamp_buffer(None)
t = buf.view(nblk, 8)[:, :6].double()
dt = (t[:, 1:] - t[:, :-1])
if d == 320:
    names = ["prologue (masks, Q frags, K0)", "phase 1 (S slab)", "phase 2 (softmax)", "phase 3 (PV)", "epilogue (store)"]
else:
    names = ["prologue (Q frags, first K/V tile)", "tile loop", "epilogue (store)"]
med = dt.median(dim=0).values
for nm, v in zip(names, med.tolist()):
    print("d=%d %-36s %9.0f cycles" % (d, nm, v))
print("total per workgroup %.0f cycles" % (t[:, len(names)] - t[:, 0]).median().item())
