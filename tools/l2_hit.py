"""L2 hit rate per kernel family from one rocprofv3 PMC pass (MI355X micro-architecture guide, section L2):
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d DIR -o l -- python3 bench.py --steps 2 --warmup 1 ...
    python3 tools/l2_hit.py DIR/l_counter_collection.csv
hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), summed over the launches of a family (tools/pmc_traffic.family names them as bench.py does)."""
import collections
import csv
import sys

from pmc_traffic import family

agg = collections.defaultdict(lambda: collections.defaultdict(float))
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = family(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(agg.items()):
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    if h + m:
        print("%-40s TCC hit %12.0f  miss %12.0f  hit rate %.3f" % (k, h, m, h / (h + m)))
