"""Diagnostic: the launch sequence of ONE KV-cached greedy step at cfg 4 (kernel names in stream order with durations and the idle gap before
each), from torch.profiler's device trace.  Usage (GPU box): python tools/decode_step_sequence.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
batch_size = sys.argv[1] if len(sys.argv) > 1 else "256"
sys.argv = ["bench.py", "--mode", "decode", "--batch", batch_size]
import bench  # noqa: E402
import case_rg_amd  # noqa: E402
from case_rg_amd.CaSE.Model import CaSE  # noqa: E402
from case_rg_amd.common.CumulativeTrainer import init_params  # noqa: E402
from case_rg_amd.common.Utils import init_seed  # noqa: E402
from case_rg_amd.utils import make_vocab, synth_batch  # noqa: E402

a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
case_rg_amd.set_compute_dtype(torch.bfloat16)
init_seed(123456)
v2i, i2v = make_vocab(a.vocab)
model = CaSE(4, 8, i2v, v2i, a.hidden, enc_layers=a.enc_layers)
init_params(model)
model = model.to(dev).eval()
batch = {k: v.to(dev) for k, v in synth_batch(a.batch, a.passages, a.passage_len, a.query_len, a.answer_len, a.vocab, seed=123456, ragged=False).items()}
with torch.no_grad():
    for _ in range(2):
        model(dict(batch), method="test")
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with torch.no_grad():
        model(dict(batch), method="test")
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
names = [e.name for e in ev]
# a cached step ends with the pointer head; take the kernels between the last two of them
marks = [i for i, n in enumerate(names) if "pointer_head_decode" in n]
if len(marks) < 3:
    print("no pointer_head_decode marks (%d kernels)" % len(ev))
    sys.exit(0)
lo, hi = marks[-3] + 1, marks[-2] + 1
prev_end = ev[lo - 1].time_range.end
tot_k = tot_gap = 0.0
for i in range(lo, hi):
    e = ev[i]
    gap = e.time_range.start - prev_end
    dur = e.time_range.end - e.time_range.start
    tot_k += dur
    tot_gap += max(0.0, gap)
    print("%4d  gap %6.1f us  run %7.1f us  %s" % (i - lo, gap, dur, e.name[:110]))
    prev_end = e.time_range.end
print("one cached step: %d launches, kernels %.1f us, gaps %.1f us, span %.1f us" % (hi - lo, tot_k, tot_gap, ev[hi - 1].time_range.end - ev[lo - 1].time_range.end))
