set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r03r
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
# keep the merge small: drop the raw traces
rm -rf $O/stats $O/enc $O/pmc_f $O/pmc_w $O/pmc_m
python3 tools/chain_stamps.py > $O/chain_stamps.txt 2>&1 || true
ls -la $O
