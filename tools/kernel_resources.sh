#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip>  -> one line per kernel: VGPR / AGPR / scratch / occupancy / spills
cd "$(dirname "$0")/../case_rg_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -ffp-contract=fast -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kr_$$.o 2>&1 |
  python3 -c '
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line.rstrip())
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:70]
    print("%-72s vgpr %4s agpr %4s scratch %4s occ %s spill %s" % (name, r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill")))
'
rm -f /tmp/kr_$$.o
