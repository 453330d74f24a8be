"""Diagnostic (GPU box): where does the bf16 gradient error of a TransformerBlock come from?  Same block, same inputs, HIP bf16 vs
the f32 CPU oracle, with the block's activation ReLU (the reference default: discontinuous derivative) and GELU (smooth)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import case_rg_amd  # noqa: E402
import cases  # noqa: E402
import oracle  # noqa: E402


def run(ns, dev, act, width, dt):
    m = cases._mod(ns.TransformerBlock(8, width, 768, activation=act), 231, dev)
    x = cases._rand(232, 1, 2, 512, width).to(dev).to(dt).requires_grad_()
    valid = cases._valid(233, 2, 512, min_len=256).reshape(1, 2, 512).to(dev)
    y = m(x, valid)
    g = torch.autograd.grad(cases._probe([y]), [x, m.self_attn.in_proj_weight, m.linear1.weight, m.linear2.weight, m.norm2.weight])
    return [y.detach().float().cpu()] + [t.detach().float().cpu() for t in g]


case_rg_amd.set_dropout(False)
for width in (768, 3840):
    for act, name in ((F.relu, "relu"), (F.gelu, "gelu")):
        want = run(oracle, torch.device("cpu"), act, width, torch.float32)
        case_rg_amd.set_compute_dtype(torch.bfloat16)
        got = run(case_rg_amd.namespace(), torch.device("cuda"), act, width, torch.bfloat16)
        errs = ["%.4f" % ((a - b).norm() / b.norm()).item() for a, b in zip(got, want)]
        print("width %4d  %s   L2 err  y %s  dx %s  d_in_proj %s  d_linear1 %s  d_linear2 %s  d_norm2 %s" % tuple([width, name] + errs))
