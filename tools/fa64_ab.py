"""Interleaved A/B timing of K18 builds (tools/fa64_variant.sh) in ONE process per library is impossible with ctypes (one library per
process), so each variant runs in its own child, three rounds interleaved: `python tools/fa64_ab.py p_drop lib1 lib2 ...`."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from case_rg_amd import config, ops
p = float(sys.argv[1]); config.set_dropout(p > 0)
N, h, L, d = 320, 8, 384, 64; E = h * d
qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.5).to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
f = lambda: ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, p_drop=p)
for _ in range(5): f()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
print(json.dumps({"ms_min": round(min(ts), 4), "ms_med": round(sorted(ts)[2], 4)}))
''' % ROOT
p = sys.argv[1]
libs = sys.argv[2:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["CASE_HIP_LIB"] = l
        out = subprocess.run([sys.executable, "-c", CHILD, p], env=env, capture_output=True, text=True)
        try:
            res[l].append(json.loads(out.stdout.strip().splitlines()[-1])["ms_min"])
        except Exception:
            res[l].append(None)
            print(out.stderr[-500:])
for l in libs:
    print(os.path.basename(l), res[l])
