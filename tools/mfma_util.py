"""MFMA utilisation per kernel from one rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o m -- python3 tools/blas_ceiling.py
    python tools/mfma_util.py DIR/m_counter_collection.csv out.json
Units (MI355X micro-architecture guide): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy, summed
over the 1024 SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of active cycles, so active cycles = GRBM_GUI_ACTIVE / 8 and
utilisation = MFMA_BUSY / (1024 * GRBM_GUI_ACTIVE / 8)."""
import collections, csv, json, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = re.sub(r"\(.*$", "", re.sub(r"void |\(anonymous namespace\)::", "", r["Kernel_Name"]))
        if not any(k in name for k in ("gemm", "fa_", "fa64", "Cijk", "chain_kernel", "attn_", "scores_kernel")):
            continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[name] += 1
out = {}
for name, c in agg.items():
    if c.get("GRBM_GUI_ACTIVE"):
        util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        out[name] = {"dispatches": cnt[name], "mfma_busy_frac_of_active_cycles": round(util, 4)}
        print("%-110s n=%-4d MFMA busy %.3f" % (name[:110], cnt[name], util))
if len(sys.argv) > 2:
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/blas_ceiling.py / bench.py (see tools/mfma_util.py)",
               "kernels": out}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
