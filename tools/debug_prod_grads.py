"""Diagnostic (GPU box): per-parameter gradient error of the fp32 HIP CaSE vs the CPU oracle at the production-shape fixture."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import case_rg_amd  # noqa: E402
import cases  # noqa: E402
import oracle  # noqa: E402


def run(ns, dev):
    m = cases._prod_model(ns, dev, 211, "case")
    b = cases._prod_batch(dev, 212, "case")
    losses = m(dict(b), method="train")
    sum(l.mean() for l in losses).backward()
    return [l.item() for l in losses], {n: p.grad.detach().cpu().double() for n, p in m.named_parameters()}


case_rg_amd.set_compute_dtype(torch.float32)
case_rg_amd.set_dropout(False)
lw, gw = run(oracle, torch.device("cpu"))
lg, gg = run(case_rg_amd.namespace(), torch.device("cuda"))
print("losses", lw, lg)
rows = []
for n in gw:
    a, b = gg[n], gw[n]
    rows.append(((a - b).norm().item() / (b.norm().item() + 1e-30), (a - b).abs().max().item() / (b.abs().max().item() + 1e-30),
                 abs(a.norm().item() - b.norm().item()) / (b.norm().item() + 1e-30), n))
for r in sorted(rows, reverse=True)[:40]:
    print("l2 %.2e  max %.2e  norm %.2e  %s" % r)
