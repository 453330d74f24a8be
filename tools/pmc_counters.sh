# usage (GPU box): bash tools/pmc_counters.sh <tag> "<CTR1 CTR2 ...>" <python script and args...>  -- one rocprofv3 PMC pass, per-kernel means
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; tag=$1; ctrs=$2; shift; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/c -o c -- python3 "$@" > $O/c.out 2> $O/c.err
python3 - $O <<'PY'
import collections, csv, glob, re, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in glob.glob(O + "/c/**/c_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*$", "", re.sub(r"void |\(anonymous namespace\)::", "", r["Kernel_Name"]))[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:14]:
    print(k)
    for c, v in sorted(agg[k].items()):
        print("    %-32s %16.0f per launch (n=%d)" % (c, v / n[(k, c)], n[(k, c)]))
PY
rm -rf $O/c
