#!/bin/bash
TAG=${1:-r04q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
L=qpth_amd/libqpx_hip.so
for kc in 1 0 2 3 4 6 9; do
  echo "== K cost extra $kc" >> $OUT/summary.txt
  QPX_PF_KCOST=$kc timeout 200 python scripts/ab_bench.py $L:0 512 100 100 0 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT/summary.txt
done
cat $OUT/summary.txt
