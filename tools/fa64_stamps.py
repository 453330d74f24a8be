"""Phase anatomy of K18 (csrc/attn64.hip) from s_memtime stamps: builds a -DFA64_STAMPS copy of the library into gpurun_out/,
runs the cfg 2 self-attention forward and prints, per wave of workgroup 0 and interval, the cycles of phase work, of the counted
vmcnt wait and of the barrier wait.  `python tools/fa64_stamps.py [p_drop]` (on the GPU box)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "fa64_stamps")
os.makedirs(OUT, exist_ok=True)
csrc = os.path.join(ROOT, "case_rg_amd", "csrc")
lib = os.path.join(OUT, "libcase_hip_stamps.so")
objs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".o") and f != "attn64.o"]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                "-ffp-contract=fast", "-DFA64_STAMPS"] + os.environ.get("FA64_EXTRA", "").split() + ["-c", os.path.join(csrc, "attn64.hip"), "-o", os.path.join(OUT, "attn64_stamps.o")], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [os.path.join(OUT, "attn64_stamps.o"), "-o", lib], check=True)
os.environ["CASE_HIP_LIB"] = lib
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from case_rg_amd import _abi, config, ops  # noqa: E402

p_drop = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "bwd" else 0.0
BWD = "bwd" in sys.argv
config.set_dropout(p_drop > 0)
N, h, L, d = 320, 8, 384, 64
E = h * d
qkv = (torch.randn(N, L, 3 * E, device="cuda") * 0.5).to(torch.bfloat16)
valid = torch.ones(N, L, dtype=torch.bool, device="cuda")
if BWD:
    p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    config.set_dropout(p_drop > 0)
    qkv.requires_grad_()
    g = torch.randn(N, L, E, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, p_drop=p_drop).backward(g)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * (12 * 24 * 7))()
    fn = _abi.lib.case_attention_resident_bwd_stamps
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    print("tile: issue+S/dP+exp+dS | barrier 1 | dV dK | dQ | vm wait | barrier 2   (cycles)")
    for w in (0, 5, 8, 11):
        print("wave", w)
        for t in range(24):
            v = [buf[(w * 24 + t) * 7 + k] for k in range(7)]
            print("  T=%2d  step1 %5d  bar %5d  dvdk %5d  dq %5d  vm %5d  bar %5d   total %5d" % (
                t + 14, v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4], v[6] - v[5], v[6] - v[0]))
    tot = buf[(0 * 24 + 23) * 7 + 6] - buf[(0 * 24 + 0) * 7 + 0]
    print("24 tiles: %d cycles = %.0f per tile" % (tot, tot / 24.0))
    sys.exit(0)
for _ in range(3):
    ops.attention(qkv, qkv, qkv, 0, E, 2 * E, h, d, key_valid=valid, p_drop=p_drop)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * (12 * 36 * 3))()
fn = _abi.lib.case_attention_resident_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
names = ["QK", "SM", "PV"]
print("interval  wave: work / vmwait / barrier   (cycles; phase of group 0 waves: QK, SM, PV, ...)")
for w in (0, 4, 8, 1):
    prev = None
    rows = []
    for t in range(36):
        a, b, c = (buf[(w * 36 + t) * 3 + k] for k in range(3))
        if prev is not None:
            rows.append((t, a - prev, b - a, c - b))
        prev = c
    grp = w // 4
    print("wave", w)
    for t, work, vm, bar in rows:
        print("  t=%2d %s  work %5d  vm %5d  barrier %5d" % (t, names[(t + 26 - grp) % 3], work, vm, bar))
tot = buf[(0 * 36 + 35) * 3 + 2] - buf[(0 * 36 + 0) * 3 + 2]
print("35 intervals: %d cycles = %.0f per interval" % (tot, tot / 35.0))
