#!/bin/bash
# Round 4, visit r (short): matrix-core pre-factorisation vs sweep over the sizes the dispatcher would give it
TAG=${1:-r04r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
L=qpth_amd/libqpx_hip.so
for dims in "512 70 50 0" "512 80 100 0" "512 96 96 0" "512 49 60 0" "512 56 112 0" "512 100 10 0" "512 112 96 0" "2048 100 100 0" "4096 64 64 0" "65536 64 64 0" "64 100 100 0"; do
  echo "== $dims" >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py $L:16384 $L:0 $dims 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT/summary.txt
done
cat $OUT/summary.txt
