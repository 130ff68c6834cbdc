#!/bin/bash
# Round 4, visit t (short): the backward with both rX products at once, against the previous build on the same box
TAG=${1:-r04t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
for dims in "512 100 100 0" "512 100 50 10" "8192 64 64 0"; do
  echo "== $dims" >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_prev.so qpth_amd/libqpx_hip.so $dims 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/summary.txt
done
timeout 200 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": {[^}]*}' >> $OUT/summary.txt
echo "(parity: the final visit)" >> $OUT/summary.txt

cat $OUT/summary.txt
