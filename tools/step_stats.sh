# usage (on the GPU box): bash tools/step_stats.sh <tag> [extra bench flags]  -- rocprofv3 kernel-stats table of the train step alone
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --no-cpu-baseline --no-north-star --no-decode-point --steps 6 --warmup 2 "$@" > $O/bench.json 2> $O/bench.err
cp $(find $O/stats -name 's_kernel_stats.csv') $O/kernel_stats.csv
rm -rf $O/stats
python3 tools/kstats.py $O/kernel_stats.csv 9 70 > $O/table.txt
tail -c 300 $O/bench.json
