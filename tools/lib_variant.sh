#!/bin/bash
# Build a copy of libcase_hip.so in which the listed sources are compiled with extra flags (A/B measurements on one box):
#   tools/lib_variant.sh NAME "-DFLAG" attn_mqa.hip attn_pointer.hip  ->  build/variants/libcase_hip_NAME.so   (use with CASE_HIP_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/variants
mkdir -p $OUT
CS=$ROOT/case_rg_amd/csrc
NAME=$1; FLAGS=$2; shift 2
SKIP=""; NEW=""
for f in "$@"; do
  b=${f%.hip}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -ffp-contract=fast $FLAGS -c $CS/$f -o $OUT/${b}_$NAME.o
  SKIP="$SKIP|/$b.o"; NEW="$NEW $OUT/${b}_$NAME.o"
done
OBJS=$(ls $CS/*.o | grep -v -E "${SKIP#|}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $NEW -o $OUT/libcase_hip_$NAME.so
echo $OUT/libcase_hip_$NAME.so
