"""Only the cfg 5 decoder cross-attention (4 items x 40 queries over the 20 480-token memory, head_dim 96): the launch whose K / V stream
bench.py --mode cfg5 reports as its HBM roofline.  Profiled with two rocprofv3 PMC passes for `traffic` (tools/pmc_traffic.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from case_rg_amd import ops  # noqa: E402

N, h, T, S, d = 4, 8, 40, 20480, 96
E = h * d
q = (torch.randn(N, T, E, device="cuda") * 0.5).to(torch.bfloat16)
kv = (torch.randn(N, S, 2 * E, device="cuda") * 0.5).to(torch.bfloat16)
valid = torch.ones(N, S, dtype=torch.bool, device="cuda")
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        ops.attention(q, kv, kv, 0, 0, E, h, d, key_valid=valid)
torch.cuda.synchronize()
