# usage (on the GPU box): bash tools/prof_step.sh <tag>  -- A/B of the K17 switch, then a rocprofv3 kernel-stats table of the train step
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for z in 1 0 1 0; do CASE_SCORES_FUSED=$z python3 bench.py --no-cpu-baseline --no-north-star --steps 12 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k17', $z, d['ms_per_step'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --no-cpu-baseline --no-north-star --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err
cp $(find $O/stats -name 's_kernel_stats.csv') $O/kernel_stats.csv
rm -rf $O/stats
python3 tools/kstats.py $O/kernel_stats.csv 8 40
