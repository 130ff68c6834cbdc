#!/bin/bash
# Round 3, visit a: probes before the loop-kernel redesign.  (1) wave -> SIMD placement, pivot-chain contention;
# (2) phase / panel profile of the round-2 loop kernel with two QPs per CU (B = 512) and one (B = 256);
# (3) FETCH_SIZE calibration of the 8-B-per-lane buffer loads.
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== probes ($(el))" | tee $OUT/summary.txt
timeout 120 python scripts/probe_simd.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_simd.txt >> $OUT/summary.txt
for B in 512 256; do
  echo "== phases B=$B ($(el))" | tee -a $OUT/summary.txt
  timeout 100 python scripts/prof_phases.py $B 100 100 0 2>&1 | grep -v amdgpu.ids | tee $OUT/phases_$B.txt >> $OUT/summary.txt
  timeout 100 python scripts/prof_panel.py $B 100 100 0 2>&1 | grep -v amdgpu.ids | tee $OUT/panel_$B.txt >> $OUT/summary.txt
done
echo "== loop kernel time at B = 512 / 256 / 128 (same kernel, fewer QPs per CU) ($(el))" | tee -a $OUT/summary.txt
for dims in "512 100 100 0" "256 100 100 0" "128 100 100 0"; do
  timeout 60 python scripts/ab_bench.py qpth_amd/libqpx_hip.so $dims 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/B n m q = $dims: /" >> $OUT/summary.txt
done
echo "== FETCH_SIZE calibration ($(el))" | tee -a $OUT/summary.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/calib -o calib -- python $REPO/scripts/calib_fetch.py > $REPO/$OUT/calib.log 2>&1); echo "rocprofv3 exit $?" >> $OUT/summary.txt
tail -2 $OUT/calib.log >> $OUT/summary.txt
find /tmp/calib -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done > $OUT/calib_fetch.txt 2>&1
grep -i "stream\|FETCH" $OUT/calib_fetch.txt | head -20 >> $OUT/summary.txt
echo "== done ($(el))" | tee -a $OUT/summary.txt
