#!/bin/bash
# Build a copy of libcase_hip.so whose attn64.o is compiled with extra flags (A/B measurements of K18 on one box):
#   tools/fa64_variant.sh NAME "-DFA64_PRIO_QK=0 -DFA64_PRIO_SM=0 -DFA64_PRIO_PV=0"  ->  gpurun_out/variants/libcase_hip_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/variants
mkdir -p $OUT
CS=$ROOT/case_rg_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -ffp-contract=fast $2 -c $CS/attn64.hip -o $OUT/attn64_$1.o
OBJS=$(ls $CS/*.o | grep -v attn64.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/attn64_$1.o -o $OUT/libcase_hip_$1.so
echo $OUT/libcase_hip_$1.so
