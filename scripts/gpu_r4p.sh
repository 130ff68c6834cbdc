#!/bin/bash
TAG=${1:-r04p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
timeout 200 python scripts/prof_prefac.py 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
timeout 200 python scripts/prof_prefac.py 256 100 100 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
L=qpth_amd/libqpx_hip.so
for dims in "512 100 100 0" "8192 64 64 0"; do
  echo "== $dims" >> $OUT/summary.txt
  timeout 200 python scripts/ab_bench.py $L:16384 $L:0 $dims 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/summary.txt
done
cat $OUT/summary.txt
