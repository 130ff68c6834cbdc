#!/bin/bash
# A/B builds of the tile translation unit (k9): libqpx_hip_<tag>.so = the shipped objects + k9 compiled with extra flags.
#   scripts/build_variants.sh tag1 "flags1" tag2 "flags2" ...
set -e
cd "$(dirname "$0")/../qpth_amd/csrc"
make -s -j"$(nproc)" all
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-variable -DQPX_TU_KERNEL=9 -DQPX_TU_REAL=double"
OBJS=$(ls _build/*.o | grep -v k9_double)
while [ $# -ge 2 ]; do
  TAG=$1; FL=$2; shift 2
  ( $HIPCC $BASE $FL -c qpx_hip_kernels.hip -o _build/k9v_$TAG.obj && $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libqpx_hip_$TAG.so $OBJS _build/k9v_$TAG.obj && echo "built libqpx_hip_$TAG.so" ) &
done
wait
