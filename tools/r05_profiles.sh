# Round-5 evidence run (GPU box): bash tools/r05_profiles.sh [a|b|all]  -> gpurun_out/r05z/*  (copy what is judged into profiles/r05_*)
# a = the default command's kernel table + PMC passes, b = the other modes (each half fits one 20-minute gpurun call)
# Every rocprofv3 command has the program itself (python3 ...) directly after `--`; counters are collected in their own passes.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05z
mkdir -p $O
cd $R
PART=${1:-all}
if [ "$PART" != b ]; then
# 1. the default bench command under kernel-trace stats, and without the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_stats.err
cp $(find $O/stats -name 's_kernel_stats.csv') $O/bench_kernel_stats.csv; rm -rf $O/stats
echo stats done
python3 bench.py > $O/bench.json 2> $O/bench.err
echo bench done
# 2. encoder mode (north-star path) and decode mode kernel tables
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o e -- python3 bench.py --mode encoder --batch 64 > $O/enc6_under_rocprof.json 2> $O/enc_stats.err
cp $(find $O/enc -name 'e_kernel_stats.csv') $O/enc6_kernel_stats.csv; rm -rf $O/enc
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -o d -- python3 bench.py --mode decode --batch 256 --no-cpu-baseline --steps 3 --warmup 1 > $O/dec_under_rocprof.json 2> $O/dec_stats.err
cp $(find $O/dec -name 'd_kernel_stats.csv') $O/dec_kernel_stats.csv; rm -rf $O/dec
echo enc dec done
# 3. PMC passes (own runs): HBM traffic, MFMA busy, L2 hit rate
W=case/b32/h512/p10x384/enc6/bf16
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_f.json 2> $O/pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_w.json 2> $O/pmc_w.err
echo write done
python3 tools/pmc_traffic.py $(find $O/pmc_f -name 'f_counter_collection.csv') $(find $O/pmc_w -name 'w_counter_collection.csv') $O/pmc_traffic.json "$W" "$CMD" > $O/pmc_traffic.txt
rm -rf $O/pmc_f $O/pmc_w
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_m.json 2> $O/pmc_m.err
python3 tools/mfma_util.py $(find $O/pmc_m -name 'm_counter_collection.csv') $O/mfma_util_step.json > $O/mfma_util.txt
rm -rf $O/pmc_m
echo mfma done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_l2 -o l -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star > $O/pmc_l2.json 2> $O/pmc_l2.err
PYTHONPATH=tools python3 tools/l2_hit.py $(find $O/pmc_l2 -name 'l_counter_collection.csv') > $O/l2_hit.txt
rm -rf $O/pmc_l2
echo l2 done
# decode-step traffic of K21 / K22 (PMC)
WD=case/b256/h512/p10x384/enc6/bf16
CD="python3 bench.py --mode decode --batch 256 --no-cpu-baseline --steps 1 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_df -o f -- python3 bench.py --mode decode --batch 256 --no-cpu-baseline --steps 1 --warmup 1 > $O/pmc_df.json 2> $O/pmc_df.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_dw -o w -- python3 bench.py --mode decode --batch 256 --no-cpu-baseline --steps 1 --warmup 1 > $O/pmc_dw.json 2> $O/pmc_dw.err
python3 tools/pmc_traffic.py $(find $O/pmc_df -name 'f_counter_collection.csv') $(find $O/pmc_dw -name 'w_counter_collection.csv') $O/dec_pmc_traffic.json "$WD" "$CD" > $O/dec_pmc_traffic.txt
rm -rf $O/pmc_df $O/pmc_dw
echo dec pmc done
fi
if [ "$PART" != a ]; then
# 4. the other modes
python3 bench.py --mode decode --batch 256 > $O/dec.json 2> $O/dec.err
python3 bench.py --mode decode --batch 256 --graph --no-cpu-baseline > $O/dec_graph.json 2> $O/dec_graph.err
CASE_DECODE_ABSORB=off CASE_POINTER_FUSED=off CASE_POINTER_HEAD=off python3 bench.py --mode decode --batch 256 --no-cpu-baseline > $O/dec_round4_path.json 2> $O/dec_round4_path.err
python3 bench.py --model masque --no-cpu-baseline --no-north-star > $O/masque.json 2> $O/masque.err
python3 bench.py --model masque --batch 8 --no-north-star --no-cpu-baseline > $O/masque_b8.json 2> $O/masque_b8.err
python3 bench.py --mode cfg5 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err
python3 bench.py --mode refdefault > $O/refdefault.json 2> $O/refdefault.err
python3 bench.py --mode encoder --batch 64 --enc-layers 3 > $O/enc3.json 2> $O/enc3.err
python3 bench.py --mode encoder --batch 64 > $O/enc6.json 2> $O/enc6.err
echo modes done
fi
ls -la $O
