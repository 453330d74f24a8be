"""Smallest call of K18 (one workgroup per item, no dropout by default): prints the error against the f32 reference."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from case_rg_amd import config, ops
N, h, L, d = (int(x) for x in (sys.argv[1:4] + ["64"])) if len(sys.argv) > 3 else (2, 8, 384, 64)
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
E = h * d
config.set_dropout(p > 0)
qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.7).to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
print("launch", N, h, L, p, flush=True)
o = ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, p_drop=p)
torch.cuda.synchronize()
q, k, v = qkv.float().split(E, dim=-1)
qh, kh, vh = [t.reshape(N, L, h, d).transpose(1, 2) for t in (q, k, v)]
ref = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), -1) @ vh).transpose(1, 2).reshape(N, L, E)
print("rel err", ((o.float() - ref).norm() / ref.norm()).item(), "finite", bool(torch.isfinite(o.float()).all()), flush=True)
