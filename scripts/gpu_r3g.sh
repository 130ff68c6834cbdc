#!/bin/bash
# Round 3: A/B of the current build against the round-2 library + phase profile
TAG=${1:-r03g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== A/B B=512 / 256" | tee $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so 256 100 100 0 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/summary.txt
for B in 256 512; do
  timeout 100 python scripts/prof_phases.py $B 100 100 0 2>&1 | grep -v amdgpu.ids | grep -v "k_sweep\|load Q\|G^T 1\|sweep n+q\|scatter" >> $OUT/summary.txt
done
