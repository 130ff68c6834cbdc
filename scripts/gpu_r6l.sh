#!/bin/bash
# crossover of the one-wave-per-QP form (variant 2048) against the chain-wave form (variant 8192) at four tile rows
TAG=${1:-r06l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for dims in "512 64 64 0" "1024 64 64 0" "1536 64 64 0" "2048 64 64 0" "4096 64 64 0" "8192 64 64 0" "512 100 50 10" "1024 100 50 10" "2048 100 50 10" "4096 100 50 10"; do
  echo "-- B n m q = $dims   (:2048 = one wave per QP, :8192 = chain-wave form)" >> $OUT/${TAG}_ab.txt
  timeout 400 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:2048 qpth_amd/libqpx_hip.so:8192 $dims 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/${TAG}_ab.txt
done
cat $OUT/${TAG}_ab.txt
