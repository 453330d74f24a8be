# usage (GPU box): bash tools/prof_mode.sh <tag> <bench.py args...>  -- rocprofv3 kernel-stats table of one bench mode
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py "$@" > $O/bench.json 2> $O/bench.err
cp $(find $O/stats -name 's_kernel_stats.csv') $O/kernel_stats.csv
rm -rf $O/stats
python3 tools/kstats.py $O/kernel_stats.csv 1 45
