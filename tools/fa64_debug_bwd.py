"""Gradient errors of K19 per output and head (debugging aid)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from case_rg_amd import config, ops
N, h, L, d = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 64) if len(sys.argv) > 3 else (2, 8, 384, 64)
E = h * d
g0 = torch.Generator().manual_seed(3)
qkv = (torch.randn(N, L, 3 * E, generator=g0) * 0.7).cuda().to(torch.bfloat16).requires_grad_()
g = torch.randn(N, L, E, generator=g0).cuda().to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid).backward(g)
r = qkv.detach().float().requires_grad_()
q, k, v = r.split(E, dim=-1)
qh, kh, vh = [t.reshape(N, L, h, d).transpose(1, 2) for t in (q, k, v)]
(torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), -1) @ vh).transpose(1, 2).reshape(N, L, E).backward(g.float())
got, ref = qkv.grad.float(), r.grad
for name, o in (("dq", 0), ("dk", E), ("dv", 2 * E)):
    a, b = got[..., o:o + E], ref[..., o:o + E]
    print(name, "rel err %.4f" % ((a - b).norm() / b.norm()).item(), "norms", a.norm().item(), b.norm().item(),
          "cos %.4f" % (torch.dot(a.flatten(), b.flatten()) / (a.norm() * b.norm())).item())
a, b = got[0, :, :64], ref[0, :, :64]   # dq of sequence 0, head 0
print("dq[0, :4, :8] got\n", a[:4, :8], "\nref\n", b[:4, :8])
for blk in range(4):
    x, y = a[:, 16 * blk:16 * blk + 16], b[:, 16 * blk:16 * blk + 16]
    print("dq head 0 d-block", blk, "err %.4f" % ((x - y).norm() / y.norm()).item())
for qb in range(12):
    x, y = a[32 * qb:32 * qb + 32], b[32 * qb:32 * qb + 32]
    print("dq head 0 tile", qb, "err %.4f" % ((x - y).norm() / y.norm()).item())
