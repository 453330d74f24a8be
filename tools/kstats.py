"""Per-step kernel time table from a rocprofv3 --kernel-trace --stats CSV: python tools/kstats.py <kernel_stats.csv> <steps incl. warm-up + instrumented> [top]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms/step %.2f over %d launches/step" % (tot / 1e6 / steps, sum(int(r['Calls']) for r in rows) / steps))
for r in rows[:top]:
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r['Name'])
    n = re.sub(r"\(.*$", "", n)[:100]
    print("%-100s %6.1f/step %8.3f ms/step  %8.1f us" % (n, int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3))
