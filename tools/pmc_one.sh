# usage (GPU box): bash tools/pmc_one.sh <tag> <python script and args...>  -- FETCH_SIZE / WRITE_SIZE per kernel of one script (two passes)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o f -- python3 "$@" > $O/f.out 2> $O/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o w -- python3 "$@" > $O/w.out 2> $O/w.err
python3 tools/pmc_table.py $O
