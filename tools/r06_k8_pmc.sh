# round 6 (GPU box): HBM-side traffic of the K8 kernels at B = 256 (two rocprofv3 PMC passes over the greedy pass's encode phase)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_k8pmc; mkdir -p $O; cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o c -- python3 $R/bench.py --mode decode --batch 256 --decode-len 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/$c.json 2> $O/$c.err)
done
python3 tools/pmc_traffic.py $(find $O/FETCH_SIZE -name 'c_counter_collection.csv') $(find $O/WRITE_SIZE -name 'c_counter_collection.csv') $O/k8_pmc_traffic.json "case/b256/h512/p10x384/enc6/bf16/decode-encode" "python3 bench.py --mode decode --batch 256 --decode-len 1 --steps 1 --warmup 1 --no-cpu-baseline" > $O/k8_pmc_traffic.txt
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
grep -E "k8|fas_fwd|chain" $O/k8_pmc_traffic.txt || true
