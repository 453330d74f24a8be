# usage (GPU box): bash tools/pmc_any.sh <tag> "<counters of pass 1>;<counters of pass 2>;..." <python script and args...>
# One rocprofv3 --pmc pass per ';'-separated group (kernel trace only), then a per-kernel table: counter totals per launch.
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; tag=$1; groups=$2; shift; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
i=0
IFS=';' read -ra G <<< "$groups"
for g in "${G[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/p$i -o p -- python3 "$@" > $O/p$i.out 2> $O/p$i.err
  i=$((i+1))
done
python3 - $O <<'PY'
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for p in glob.glob(sys.argv[1] + "/p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*$", "", re.sub(r"void |\(anonymous namespace\)::", "", r["Kernel_Name"]))[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in agg.items():
    print(k)
    for name in sorted(c):
        print("    %-32s %.4e per launch (n=%d)" % (name, c[name] / n[k][name], n[k][name]))
PY
