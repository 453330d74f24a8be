"""Reads the s_memtime stamps a -DCHAIN_STAMPS build of csrc/encoder_chain.hip leaves in the tail of qkv_out (diagnostic build only,
CASE_HIP_LIB=...): per-phase cycles of wave STAMP_WAVE on each workgroup's second tile, median over workgroups.
Stamp ids: 0 tile start, 1 own DMA rows landed, 2 barrier passed, then per stage st: 3+4st K loop done, 4+4st barrier passed (stages 0-2),
5+4st epilogue done, 6+4st closing barrier passed; 27 tile end."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import case_rg_amd  # noqa: E402
from case_rg_amd import ops  # noqa: E402
from case_rg_amd.utils import fill_params  # noqa: E402

dev = torch.device("cuda", 0)
case_rg_amd.set_compute_dtype(torch.bfloat16)
ns = case_rg_amd.namespace()
layer = ns.TransformerEncoderLayer(512, 8, dim_feedforward=512, dropout=0.1, activation="gelu")
enc = fill_params(ns.TransformerEncoder(layer, 2), 3, gain=2.0).to(dev).eval()
N, L = 640, 384
x = torch.randn(N, L, 512, device=dev).to(torch.bfloat16)
s = torch.randn(N, L, 512, device=dev).to(torch.bfloat16)
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
with torch.no_grad():
    for _ in range(3):
        if variant == "head":
            s_out, qkv = ops.encoder_chain("head", x, None, None, enc.layers[0])
        elif variant == "tail":
            s_out, qkv = ops.encoder_chain("tail", x, s, enc.layers[1], None)
        else:
            s_out, qkv = ops.encoder_chain("full", x, s, enc.layers[0], enc.layers[1])
    torch.cuda.synchronize()
import ctypes
from case_rg_amd import _abi
buf = (ctypes.c_uint64 * (256 * 32))()
assert _abi.lib.case_encoder_chain_stamps(buf) == 0
raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(256, 32)
names = {0: "start", 1: "dma landed", 2: "barrier", 27: "tile end"}
for st in range(6):
    names[3 + 4 * st] = "st%d K loop" % st
    names[4 + 4 * st] = "st%d barrier" % st
    names[5 + 4 * st] = "st%d epilogue" % st
    names[6 + 4 * st] = "st%d closing barrier" % st
ids = [i for i in range(28) if np.median(raw[:, i]) > 0 and i in names]
prev = None
total = 0
for i in ids:
    t = raw[:, i].astype(np.float64)
    if prev is not None:
        d = np.median(t - prev)
        total += d
        print("%-22s %9.0f cycles" % (names[i], d))
    prev = t
print("sum", total)
