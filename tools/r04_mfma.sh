# Round-4 MFMA-busy pass of the default bench command (GPU box): bash tools/r04_mfma.sh -> gpurun_out/r04m/
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_m.json 2> $O/pmc_m.err
python3 tools/mfma_util.py $(find $O/pmc_m -name 'm_counter_collection.csv') $O/mfma_util_step.json > $O/mfma_util.txt
rm -rf $O/pmc_m
