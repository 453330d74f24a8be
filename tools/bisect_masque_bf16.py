"""where does the bf16_auto gradient error of prod_masque_train come from (0.148 on
response_generation.decoder.attns.1.linear_key.weight, 0.107 on query_encoder.embedding.0.weight)?
Runs the production-shape Masque fixture's model on the CPU oracle (f32) and on the product in bf16_auto, prints the FULL-tensor
relative L2 error / cosine of every parameter gradient (the fixtures hold strided slices), then repeats the bf16 run with single
switches thrown: key projection of the additive attention kept in f32, the additive-attention gradient d_uh kept in f32, attention
resident kernels off, 128 x 128 GEMM tiling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import case_rg_amd  # noqa: E402
import cases  # noqa: E402
import oracle  # noqa: E402
from case_rg_amd import ops  # noqa: E402
from case_rg_amd.common import BilinearAttention as BA  # noqa: E402

MODEL = sys.argv[1] if len(sys.argv) > 1 else "masque"
SEED = 221 if MODEL == "masque" else 211
WATCH = ["response_generation.decoder.attns.1.linear_key.weight", "response_generation.decoder.attns.0.linear_key.weight",
         "query_encoder.embedding.0.weight", "response_generation.decoder.attns.1.v.weight",
         "response_generation.decoder.attns.1.linear_query.weight"]


def run(ns, dev, dtype):
    case_rg_amd.set_compute_dtype(dtype)
    case_rg_amd.set_dropout(False)
    if hasattr(ns, "act_dtype"):
        ns.act_dtype = dtype
    m = cases._prod_model(ns, dev, SEED, MODEL)
    b = cases._prod_batch(dev, SEED + 1, MODEL)
    losses = m(dict(b), method="train")
    sum(l.mean() for l in losses).backward()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    case_rg_amd.set_compute_dtype(torch.float32)
    return {n: p.grad.detach().cpu().double() for n, p in m.named_parameters() if p.grad is not None}


def report(tag, got, want, top=8):
    rows = []
    for n in want:
        a, b = got[n], want[n]
        l2 = (a - b).norm().item() / (b.norm().item() + 1e-30)
        cos = torch.dot(a.flatten(), b.flatten()).item() / (a.norm().item() * b.norm().item() + 1e-30)
        rows.append((l2, cos, b.norm().item(), n))
    rows.sort(reverse=True)
    print("==", tag, " mean l2 %.4f" % (sum(r[0] for r in rows) / len(rows)))
    for r in rows[:top]:
        print("   l2 %.4f cos %.5f |g| %.3e  %s" % r)
    for w in WATCH:
        for r in rows:
            if r[3] == w:
                print("   [watch] l2 %.4f cos %.5f |g| %.3e  %s" % r)


want = run(oracle, torch.device("cpu"), torch.float32)
dev = torch.device("cuda")
report("fp32 product", run(case_rg_amd.namespace(), dev, torch.float32), want, top=3)
report("bf16_auto", run(case_rg_amd.namespace(), dev, torch.bfloat16), want)

# switch 1: uh = Wk k of the additive attention kept in f32 (then d_uh stays f32 too)
keep = BA.BilinearAttention.project_keys
BA.BilinearAttention.project_keys = lambda self, key: ops.linear(key, self.linear_key.weight, out_dtype=torch.float32)
try:
    report("bf16 + additive-attention keys projected to f32", run(case_rg_amd.namespace(), dev, torch.bfloat16), want)
finally:
    BA.BilinearAttention.project_keys = keep

# switch 2: 128 x 128 GEMM tiling only
ops.GEMM_TILE = 128
try:
    report("bf16 + 128 x 128 GEMM tiling", run(case_rg_amd.namespace(), dev, torch.bfloat16), want)
finally:
    ops.GEMM_TILE = 0

# switch 3: unfused attention
ops.ATTENTION_MODE = "unfused"
try:
    report("bf16 + unfused attention", run(case_rg_amd.namespace(), dev, torch.bfloat16), want)
finally:
    ops.ATTENTION_MODE = "auto"
