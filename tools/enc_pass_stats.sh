# round 6 (GPU box): kernel table of the greedy pass's ENCODE phase alone (B = 256, one-token answers: encode + memory projections + one step)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_encpass; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 $R/bench.py --mode decode --batch 256 --decode-len 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err)
cp $(find $O/stats -name 's_kernel_stats.csv') $O/kernel_stats.csv; rm -rf $O/stats
cd $R && python3 tools/kstats.py $O/kernel_stats.csv 9 60 > $O/table.txt; head -45 $O/table.txt
