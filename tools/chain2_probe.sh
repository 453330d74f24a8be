# round-6 experiment driver (GPU box): two-context encoder chain, workgroups per CU + kernel table
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in 1 2 3; do
  CASE_CHAIN_TWO_CTX=1 CASE_CHAIN_STAGGER=0 CASE_CHAIN_TWO_CTX_WGS=$w python3 bench.py --mode encoder --batch 64 --steps 20 --warmup 3 2>/dev/null > /tmp/o.json
  python3 -c "import json; d=json.load(open('/tmp/o.json')); print('wgs_per_cu=$w', d['ms_per_step'], d['roofline']['frac'])"
done
export TMPDIR=/tmp
export CASE_CHAIN_TWO_CTX=1 CASE_CHAIN_STAGGER=0
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c2prof -o c -- python3 $R/bench.py --mode encoder --batch 64 --steps 10 --warmup 2 > /dev/null 2>&1)
python3 tools/kstats.py $(find gpurun_out/c2prof -name "c_kernel_stats.csv") 12 12 | head -14
rm -rf gpurun_out/c2prof
