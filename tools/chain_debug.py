"""Compares each variant of the fused encoder chain with the same arithmetic done by torch on the GPU (f32 from the bf16 inputs); prints the
worst error and where non-finite values sit (token rows / feature columns)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import case_rg_amd  # noqa: E402
from case_rg_amd import ops  # noqa: E402
from case_rg_amd.utils import fill_params  # noqa: E402

dev = torch.device("cuda", 0)
case_rg_amd.set_compute_dtype(torch.bfloat16)
ns = case_rg_amd.namespace()
layer = ns.TransformerEncoderLayer(512, 8, dim_feedforward=512, dropout=0.1, activation="gelu")
enc = fill_params(ns.TransformerEncoder(layer, 2), 3, gain=2.0).to(dev).eval()
N, L = int(sys.argv[1]) if len(sys.argv) > 1 else 3, 384
torch.manual_seed(0)
x = torch.randn(N, L, 512, device=dev).to(torch.bfloat16)
s = torch.randn(N, L, 512, device=dev).to(torch.bfloat16)
l0, l1 = enc.layers


def ref_full(ctx, s, lay, nxt):
    bf = lambda t: t.to(torch.bfloat16).float()
    W = lambda p: bf(p.detach())
    y = ctx.float() @ W(lay.self_attn.out_proj.weight).t() + lay.self_attn.out_proj.bias + s.float()
    s2 = bf(F.layer_norm(y, (512,), lay.norm2.weight, lay.norm2.bias, lay.norm2.eps))
    a = bf(F.gelu(s2 @ W(lay.linear1.weight).t() + lay.linear1.bias))
    o = a @ W(lay.linear2.weight).t() + lay.linear2.bias + s2
    if nxt is None:
        return o, None
    sp = bf(F.layer_norm(o, (512,), nxt.norm1.weight, nxt.norm1.bias, nxt.norm1.eps))
    return sp, sp @ W(nxt.self_attn.in_proj_weight).t() + nxt.self_attn.in_proj_bias


def report(name, got, want):
    got = got.float()
    bad = ~torch.isfinite(got)
    err = ((got - want).abs() * (~bad)).max().item() / want.abs().max().item()
    print("%-10s max err / scale %.3e   non-finite %d" % (name, err, int(bad.sum())))
    if bad.any():
        idx = bad.reshape(-1, got.shape[-1]).nonzero()
        rows, cols = idx[:, 0], idx[:, 1]
        print("   rows %s ... cols %s ..." % (sorted(set((rows % 128).tolist()))[:20], sorted(set(cols.tolist()))[:40]))
        print("   (row, col):", [(int(r), int(c)) for r, c in zip(rows.tolist(), cols.tolist())][:24])
        wrong = ((got - want).abs() > 0.05 * want.abs().max()) & ~bad
        print("   finite but wrong:", int(wrong.sum()))
    wrong = ((got - want).abs() > 0.05 * want.abs().max()) | bad
    if wrong.any():
        idx = wrong.reshape(-1, got.shape[-1]).nonzero()
        import collections
        hr = collections.Counter((idx[:, 0] % 128).tolist())
        hc = collections.Counter((idx[:, 1] % 64).tolist())
        hw = collections.Counter(((idx[:, 1] % 512) // 64).tolist())
        print("   wrong by row%128:", sorted(hr.items())[:40])
        print("   wrong by col%64:", sorted(hc.items()))
        print("   wrong by wave:", sorted(hw.items()), " by tile:", sorted(collections.Counter((idx[:, 0] // 128).tolist()).items())[:12])


with torch.no_grad():
  for rep in range(3):
    sp, qkv = ops.encoder_chain("full", x, s, l0, l1)
    wsp, wqkv = ref_full(x, s, l0, l1)
    report("full s'", sp, wsp)
    report("full qkv", qkv, wqkv)
  if True:
    o, _ = ops.encoder_chain("tail", x, s, l1, None)
    wo, _ = ref_full(x, s, l1, None)
    report("tail o", o, wo)
    hs, hq = ops.encoder_chain("head", x, None, None, l0)
    ws = F.layer_norm(x.float(), (512,), l0.norm1.weight, l0.norm1.bias, l0.norm1.eps).to(torch.bfloat16).float()
    report("head s", hs, ws)
    report("head qkv", hq, ws @ l0.self_attn.in_proj_weight.detach().to(torch.bfloat16).float().t() + l0.self_attn.in_proj_bias)
