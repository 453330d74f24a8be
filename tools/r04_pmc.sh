# Round-4 PMC traffic of the default bench command (GPU box): bash tools/r04_pmc.sh -> gpurun_out/r04p/pmc_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.json 2> $O/pmc_f.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.json 2> $O/pmc_w.err
echo write done
python3 tools/pmc_traffic.py $(find $O/pmc_f -name 'f_counter_collection.csv') $(find $O/pmc_w -name 'w_counter_collection.csv') $O/pmc_traffic.json
rm -rf $O/pmc_f $O/pmc_w
