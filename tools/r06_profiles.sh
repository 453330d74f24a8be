# Round-6 evidence run (GPU box): bash tools/r06_profiles.sh [a|b|all]  -> gpurun_out/r06z/*  (copy what is judged into profiles/r06_*)
# a = the default command + ONE kernel table per workload + PMC passes; b = the other modes.  Every rocprofv3 command has the program itself
# (python3 ...) directly after `--`; counters are collected in their own passes (never with a trace domain beyond --kernel-trace).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06z
mkdir -p $O
cd $R
PART=${1:-all}
stats() {  # stats <tag> <bench flags...>: rocprofv3 kernel-stats table of exactly that command
  local tag=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -o s -- python3 $R/bench.py "$@" > $O/${tag}_under_rocprof.json 2> $O/${tag}_stats.err)
  cp $(find $O/st_$tag -name 's_kernel_stats.csv') $O/${tag}_kernel_stats.csv; rm -rf $O/st_$tag
  echo "stats $tag done"
}
pmc() {  # pmc <tag> <counters> <bench flags...>
  local tag=$1 ctr=$2; shift 2
  (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$tag -o c -- python3 $R/bench.py "$@" > $O/pmc_$tag.json 2> $O/pmc_$tag.err)
}
if [ "$PART" != b ]; then
# 1. the driver's default command, without the profiler
python3 bench.py > $O/bench.json 2> $O/bench.err; echo bench done
# 2. one kernel table PER WORKLOAD: the training step alone, the encoder forward alone, the greedy pass alone, its encode phase alone
stats train --no-cpu-baseline --no-north-star --no-decode-point --steps 8 --warmup 2
stats enc6 --mode encoder --batch 64
stats dec --mode decode --batch 256 --no-cpu-baseline --steps 3 --warmup 1
stats decenc --mode decode --batch 256 --decode-len 1 --no-cpu-baseline --steps 4 --warmup 1
# 3. PMC passes on the training step alone (own runs): HBM traffic, MFMA busy, L2 hit rate
W=case/b32/h512/p10x384/enc6/bf16
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-north-star"
pmc f FETCH_SIZE --steps 2 --warmup 1 --no-cpu-baseline --no-north-star; echo fetch done
pmc w WRITE_SIZE --steps 2 --warmup 1 --no-cpu-baseline --no-north-star; echo write done
python3 tools/pmc_traffic.py $(find $O/pmc_f -name 'c_counter_collection.csv') $(find $O/pmc_w -name 'c_counter_collection.csv') $O/pmc_traffic.json "$W" "$CMD" > $O/pmc_traffic.txt
rm -rf $O/pmc_f $O/pmc_w
pmc m "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --steps 2 --warmup 1 --no-cpu-baseline --no-north-star
python3 tools/mfma_util.py $(find $O/pmc_m -name 'c_counter_collection.csv') $O/mfma_util_step.json > $O/mfma_util.txt; rm -rf $O/pmc_m; echo mfma done
pmc l2 "TCC_HIT_sum TCC_MISS_sum" --steps 2 --warmup 1 --no-cpu-baseline --no-north-star
PYTHONPATH=tools python3 tools/l2_hit.py $(find $O/pmc_l2 -name 'c_counter_collection.csv') > $O/l2_hit.txt; rm -rf $O/pmc_l2; echo l2 done
fi
if [ "$PART" != a ]; then
# 4. the other modes (captured steps: --graph)
python3 bench.py --mode refdefault --steps 30 --warmup 3 > $O/refdefault.json 2> $O/refdefault.err
python3 bench.py --mode refdefault --steps 30 --warmup 3 --graph > $O/refdefault_graph.json 2> $O/refdefault_graph.err
python3 bench.py --model masque --batch 8 --no-north-star --no-cpu-baseline > $O/masque_b8.json 2> $O/masque_b8.err
python3 bench.py --model masque --batch 8 --no-north-star --no-cpu-baseline --no-roofline --graph > $O/masque_b8_graph.json 2> $O/masque_b8_graph.err
python3 bench.py --model masque --no-cpu-baseline --no-north-star > $O/masque.json 2> $O/masque.err
python3 bench.py --no-cpu-baseline --no-north-star --graph > $O/bench_graph.json 2> $O/bench_graph.err
python3 bench.py --mode cfg5 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err
python3 bench.py --mode decode --batch 256 > $O/dec.json 2> $O/dec.err
CASE_INTERACTION_FUSED=off python3 bench.py --mode decode --batch 256 --no-cpu-baseline > $O/dec_interaction_single_launches.json 2> $O/dec_isl.err
python3 bench.py --mode decode --batch 256 --graph --no-cpu-baseline > $O/dec_graph.json 2> $O/dec_graph.err
python3 bench.py --mode encoder --batch 64 > $O/enc6.json 2> $O/enc6.err
CASE_CHAIN_TWO_CTX=1 python3 bench.py --mode encoder --batch 64 > $O/enc6_two_ctx.json 2> $O/enc6_two_ctx.err
python3 bench.py --mode encoder --batch 64 --enc-layers 3 > $O/enc3.json 2> $O/enc3.err
CASE_FORCE_GRADSYNC=1 python3 bench.py --no-cpu-baseline --no-north-star --no-roofline > $O/one_rank_rccl.json 2> $O/one_rank_rccl.err
CASE_FORCE_GRADSYNC=1 python3 bench.py --mode refdefault --steps 30 --warmup 3 --graph > $O/one_rank_rccl_refdefault_graph.json 2> $O/one_rank_rccl_refdefault_graph.err
echo modes done
fi
ls -la $O
