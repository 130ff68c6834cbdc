#!/bin/bash
# Round 3: phase profile of the loop kernel (profiling build), B = 256 (one QP per CU) and 512
TAG=${1:-r03f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for B in 256 512; do
  timeout 100 python scripts/prof_phases.py $B 100 100 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
done
