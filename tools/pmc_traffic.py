"""HBM traffic per launch of the GEMM kernels from two rocprofv3 PMC passes of bench.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv profiles/r01_d_pmc_traffic.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): both counters are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, so wide coalesced reads are DOUBLED; WRITE_SIZE is exact for 16-byte-per-lane stores and
f32 atomics.  Separate passes because FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not fit together.
"""
import collections
import csv
import json
import re
import sys


def family(name):
    """C++ kernel name -> the family key bench.py prints."""
    m = re.search(r"gemm8w_kernel<(unsigned short|float), (true|false), (true|false), (true|false|\d)>", name)  # last: 0 plain, 1 atomics, 2 slabs
    if m:
        return "gemm8w_kernel<bf16,%s,%s,%s>" % ("bf16" if m.group(1) == "unsigned short" else "f32",
                                                    "Ak" if m.group(2) == "true" else "A", "Bk" if m.group(3) == "true" else "B")
    m = re.search(r"gemm_small_kernel<(unsigned short|float), (true|false), (true|false)>", name)
    if m:
        return "gemm_small_kernel<bf16,%s,%s,%s>" % ("bf16" if m.group(1) == "unsigned short" else "f32",
                                                       "Ak" if m.group(2) == "true" else "A", "Bk" if m.group(3) == "true" else "B")
    m = re.search(r"gemm_kernel<(unsigned short|float), (unsigned short|float), (true|false), (true|false), (true|false)>", name)
    if m:
        t = {"unsigned short": "bf16", "float": "f32"}
        return "gemm_kernel<%s,%s,%s,%s>" % (t[m.group(1)], t[m.group(2)], "Ak" if m.group(3) == "true" else "A",
                                             "Bk" if m.group(4) == "true" else "B")
    m = re.search(r"(k8::products_kernel|k8::scores_kernel<\d>|fas_fwd_kernel<\d+|chain2_kernel<\d>)", name)  # round 6
    if m:
        return m.group(1)
    m = re.search(r"chain_kernel<(\d), (?:true|false)>", name)
    if m:
        return "chain_kernel<%s>" % {"0": "full", "1": "tail", "2": "head"}.get(m.group(1), m.group(1))
    m = re.search(r"chain_kernel<(\d)>", name)
    if m:
        return "chain_kernel<%s>" % {"0": "full", "1": "tail", "2": "head"}.get(m.group(1), m.group(1))
    m = re.search(r"(scores_kernel<(?:true|false)>|rc_gemm_kernel<(?:true|false)>)", name)
    if m:
        return m.group(1)
    m = re.search(r"(fa64::fwd_kernel<(?:true|false)>|fa64::bwd::bwd_kernel<(?:true|false)>|fa64::bwd::stat_kernel|splitk_reduce_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"(mqa_decode_kernel|pointer_attend_decode_kernel|pointer_head_decode_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"(attn_decode\w*|fa_fwd\w*_kernel|fa_bwd\w*_kernel)", name)
    if m:
        return m.group(1)
    return None


def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            k = family(r["Kernel_Name"])
            if k:
                agg[k][0] += float(r["Counter_Value"]) * 1024.0
                agg[k][1] += 1
    return agg


def main():
    fpath, wpath, out = sys.argv[1:4]
    # optional 4th / 5th argument: the workload key bench.py matches against (its `workload_key`, e.g. "case/b32/h512/p10x384/enc6/bf16")
    # and the profiled command as it was run; bench.py refuses a file whose workload differs from the run it is printing
    workload = sys.argv[4] if len(sys.argv) > 4 else "case/b32/h512/p10x384/enc6/bf16"
    command = sys.argv[5] if len(sys.argv) > 5 else "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"

    f, w = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fb = 2.0 * f[k][0] / max(1, f[k][1])  # gfx950 correction: FETCH_SIZE reports half of wide coalesced reads
        wb = w[k][0] / max(1, w[k][1])
        res[k] = {"fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb),
                  "launches_fetch_pass": f[k][1], "launches_write_pass": w[k][1]}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (separate passes) -- " + command, "workload": workload,
               "corrections": "KiB -> bytes; FETCH_SIZE x 2 (gfx950 counts 128-byte requests as 64 bytes)", "kernels": res},
              open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-36s fetch %8.1f MB  write %8.1f MB per launch" % (k, v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
