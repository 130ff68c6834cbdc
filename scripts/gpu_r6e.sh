#!/bin/bash
TAG=${1:-r06e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
S=$OUT/summary.txt
: > $S
for dims in "512 100 100 0" "256 100 100 0" "512 100 50 10"; do
  timeout 300 python scripts/prof_phases.py $dims 2>&1 | grep -v amdgpu.ids | grep -v "k_sweep\|load Q\|G^T 1\|sweep n\|scatter" >> $OUT/${TAG}_phases.txt
done
cat $OUT/${TAG}_phases.txt >> $S
