# Round 6, after the greedy-step changes (non-temporal streams, cache append in the attention launch, distributions for the last step only):
# the default line, the decode lines and the decode kernel table again.  bash tools/r06_decode_refresh.sh  -> gpurun_out/r06y/*
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06y
mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err; echo bench done
python3 bench.py --mode decode --batch 256 > $O/dec.json 2> $O/dec.err
python3 bench.py --mode decode --batch 256 --graph --no-cpu-baseline > $O/dec_graph.json 2> $O/dec_graph.err
CASE_DECODE_APPEND=off python3 bench.py --mode decode --batch 256 --no-cpu-baseline > $O/dec_append_off.json 2> $O/dec_append_off.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_dec -o s -- python3 $R/bench.py --mode decode --batch 256 --no-cpu-baseline --steps 3 --warmup 1 > $O/dec_under_rocprof.json 2> $O/dec_stats.err)
cp $(find $O/st_dec -name 's_kernel_stats.csv') $O/dec_kernel_stats.csv; rm -rf $O/st_dec
echo decode refresh done
