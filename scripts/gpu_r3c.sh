#!/bin/bash
# Round 3, visit c: where the chain-wave form spends a panel (per-wave interval timers), phase profile, A/B.
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== chain-form panel timers ($(el))" | tee $OUT/summary.txt
for B in 256 512; do
  timeout 100 python scripts/prof_panel.py $B 100 100 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/panel.txt >> $OUT/summary.txt
  timeout 100 python scripts/prof_phases.py $B 100 100 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/phases.txt >> $OUT/summary.txt
done
echo "== A/B ($(el))" | tee -a $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so qpth_amd/libqpx_hip.so:16384 2>&1 | grep -v amdgpu.ids | tail -3 >> $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so qpth_amd/libqpx_hip.so:16384 256 100 100 0 2>&1 | grep -v amdgpu.ids | tail -3 >> $OUT/summary.txt
echo "== done ($(el))" | tee -a $OUT/summary.txt
