"""Per-basic-block instruction census of one kernel in a hipcc -save-temps .s file:  asm_blocks.py file.s <mangled-name-substring>"""
import re
import sys

text = open(sys.argv[1]).read()
pat = sys.argv[2]
lines = text.split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], [lines[start].strip(), []]
for l in lines[start + 1:end + 1]:
    if re.match(r"^\.LBB\S+:", l):
        blocks.append(cur)
        cur = [l.strip(), []]
    else:
        cur[1].append(l.strip())
blocks.append(cur)
for name, ins in blocks:
    ops = [i.split()[0] for i in ins if i and not i.startswith((";", "."))]
    c = lambda p: sum(1 for o in ops if o.startswith(p))
    print("%-14s n=%4d mfma=%3d exp=%3d valu=%4d ds_read=%3d ds_write=%3d vmem=%3d accr=%3d accw=%3d scratch=%d barrier=%d waitcnt=%d" % (
        name[:14], len(ops), c("v_mfma"), c("v_exp"), sum(1 for o in ops if o.startswith("v_") and not o.startswith(("v_mfma", "v_accvgpr"))),
        c("ds_read"), c("ds_write"), c("global_load") + c("buffer_load"), c("v_accvgpr_read"), c("v_accvgpr_write"), c("scratch_"),
        c("s_barrier"), c("s_waitcnt")))
