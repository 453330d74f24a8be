"""Per-sequence error of K18 under a few validity patterns (debugging aid)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from case_rg_amd import ops
N, h, L, d = 4, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 320, 64
E = h * d
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(N, L, 3 * E, generator=g) * 0.7).cuda().to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
valid[1] = (torch.rand(L, generator=g) > 0.3).cuda()
valid[2, 1:] = False
valid[3, L // 2 + 3:] = False
for rep in range(2):
    o = ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid)
    torch.cuda.synchronize()
    q, k, v = qkv.float().split(E, dim=-1)
    qh, kh, vh = [t.reshape(N, L, h, d).transpose(1, 2) for t in (q, k, v)]
    s = (qh @ kh.transpose(-1, -2) / math.sqrt(d)).masked_fill(~valid[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(N, L, E)
    for n in range(N):
        e = ((o[n].float() - ref[n]).norm() / ref[n].norm()).item()
        eh = [round(((o[n, :, 64 * j:64 * j + 64].float() - ref[n, :, 64 * j:64 * j + 64]).norm() / ref[n, :, 64 * j:64 * j + 64].norm()).item(), 4) for j in range(h)]
        print("rep", rep, "seq", n, "rel err %.4f" % e, "per head", eh)
