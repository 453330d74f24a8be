"""Condensed instruction stream of a kernel's MFMA region from hipcc -S output (runs of equal opcodes folded)."""
import sys
lines = open(sys.argv[1]).read().split('\n')
idx = [i for i, l in enumerate(lines) if 'v_mfma' in l]
print(len(idx), "mfma;", "lines", idx[0], idx[-1])
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 80
seq = []
for l in lines[idx[0] - lo: idx[-1] + 5]:
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';', '//')):
        continue
    keep = t[0].startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_setprio', 's_branch')) or t[0].endswith(':')
    seq.append(t[0] + (' ' + ' '.join(t[1:]) if keep else ''))
out, prev, cnt = [], None, 0
for s in seq:
    if s == prev:
        cnt += 1
    else:
        if prev:
            out.append(prev + (' x%d' % cnt if cnt > 1 else ''))
        prev, cnt = s, 1
out.append(prev + (' x%d' % cnt if cnt > 1 else ''))
print('\n'.join(out))
