# usage (GPU box): bash tools/pmc_lds.sh <tag> <python script and args...>  -- LDS bank-conflict share and issue mix per kernel of one script
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/l -o l -- python3 "$@" > $O/l.out 2> $O/l.err
python3 - $O <<'PY'
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in glob.glob(sys.argv[1] + "/l/**/l_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*$", "", re.sub(r"void |\(anonymous namespace\)::", "", r["Kernel_Name"]))[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    print("%-70s conflict %.3e  active %.3e  conflict/active %.3f  lds insts %.3e" % (k, c["SQ_LDS_BANK_CONFLICT"], c["SQ_LDS_IDX_ACTIVE"], c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), c["SQ_INSTS_LDS"]))
PY
