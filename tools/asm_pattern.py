"""Instruction-order sketch of the MFMA-bearing basic blocks of one kernel:  asm_pattern.py file.s <name-substring> [min_mfma] [max_blocks]
M = MFMA, r / w = LDS read / write, g = global or buffer load, e = v_exp, a = v_accvgpr, . = other VALU, s = scalar, B = barrier,
|...| = s_waitcnt (L = lgkmcnt, V = vmcnt)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
max_blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 4
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], None
for l in lines[start:end]:
    t = l.strip()
    if t.startswith(".LBB"):
        cur = [t, []]
        blocks.append(cur)
        continue
    if cur is None or not t or t.startswith(";"):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"): cur[1].append("M")
    elif op.startswith("ds_read"): cur[1].append("r")
    elif op.startswith("ds_write"): cur[1].append("w")
    elif op.startswith("s_waitcnt"): cur[1].append("|" + t.split(None, 1)[1].replace("lgkmcnt", "L").replace("vmcnt", "V").replace(" ", "") + "|")
    elif op.startswith("s_barrier"): cur[1].append("B")
    elif op.startswith(("global_load", "buffer_load")): cur[1].append("g")
    elif op.startswith("v_exp"): cur[1].append("e")
    elif op.startswith("v_accvgpr"): cur[1].append("a")
    elif op.startswith("v_"): cur[1].append(".")
    else: cur[1].append("s")
shown = 0
for name, b in blocks:
    st = "".join(b)
    if st.count("M") >= min_mfma and shown < max_blocks:
        print(name, st)
        shown += 1
