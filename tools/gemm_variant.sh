#!/bin/bash
# Build a copy of libcase_hip.so whose gemm.o is compiled with extra flags (A/B measurements of the GEMM kernels on one box):
#   tools/gemm_variant.sh NAME "-DG8_TIMING_ONLY_NO_ATOMICS"  ->  build/variants/libcase_hip_NAME.so   (use with CASE_HIP_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants
mkdir -p $OUT
CS=$ROOT/case_rg_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -ffp-contract=fast $2 -c $CS/gemm.hip -o $OUT/gemm_$1.o
OBJS=$(ls $CS/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/gemm_$1.o -o $OUT/libcase_hip_$1.so
echo $OUT/libcase_hip_$1.so
