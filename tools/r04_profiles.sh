# Round-4 evidence run (GPU box): bash tools/r04_profiles.sh  -> gpurun_out/r04z/*  (copy what is judged into profiles/r04_*)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r04z
mkdir -p $O
cd $R
# 1. the default bench command under kernel-trace stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py > $O/bench_stats.json 2> $O/bench_stats.err
echo stats done
# 2. encoder mode (north-star path) stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -o e -- python3 bench.py --mode encoder --batch 64 > $O/enc_stats.json 2> $O/enc_stats.err
echo enc done
# 3. PMC passes (own runs, kernel-trace only)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.json 2> $O/pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.json 2> $O/pmc_w.err
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_m.json 2> $O/pmc_m.err
echo mfma done
python3 tools/pmc_traffic.py $(find $O/pmc_f -name 'f_counter_collection.csv') $(find $O/pmc_w -name 'w_counter_collection.csv') $O/pmc_traffic.json
python3 tools/mfma_util.py $(find $O/pmc_m -name 'm_counter_collection.csv') $O/mfma_util_step.json > $O/mfma_util.txt
cp $(find $O/stats -name 's_kernel_stats.csv') $O/bench_kernel_stats.csv
cp $(find $O/enc -name 'e_kernel_stats.csv') $O/enc_kernel_stats.csv
rm -rf $O/stats $O/enc $O/pmc_f $O/pmc_w $O/pmc_m
# 4. the other modes and the micro-benchmarks
python3 bench.py --mode decode --batch 256 > $O/dec.json 2> $O/dec.err
python3 bench.py --mode decode --batch 256 --graph --no-cpu-baseline > $O/dec_graph.json 2> $O/dec_graph.err
python3 bench.py --model masque --no-cpu-baseline --no-north-star > $O/masque.json 2> $O/masque.err
python3 bench.py --mode cfg5 > $O/cfg5.json 2> $O/cfg5.err
python3 bench.py --model masque --batch 8 --no-north-star > $O/masque_b8.json 2> $O/masque_b8.err
python3 bench.py --mode encoder --batch 64 --enc-layers 3 > $O/enc3.json 2> $O/enc3.err
python3 bench.py --mode encoder --batch 64 > $O/enc6.json 2> $O/enc6.err
python3 tools/scores_bench.py > $O/scores_bench.jsonl 2> $O/scores_bench.err
python3 tools/attn_bench.py > $O/attn_bench.jsonl 2> $O/attn_bench.err
python3 tools/chain_bench.py > $O/chain_bench.jsonl 2> $O/chain_bench.err
python3 tools/splitk_sweep.py > $O/splitk_sweep.txt 2> $O/splitk_sweep.err
ls -la $O
