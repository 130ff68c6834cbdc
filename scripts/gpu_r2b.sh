#!/bin/bash
# Round-2 second GPU visit: A/B of the re-structured tile panel against the old one on ONE box, parity tests on the new
# build, in-kernel phase/panel timers of the new build.
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== A/B C2 (B=512 n=100 m=100)" > $OUT/summary.txt
timeout 600 python scripts/ab_bench.py qpth_amd/libqpx_hip_old.so qpth_amd/libqpx_hip.so qpth_amd/libqpx_hip_nofence.so qpth_amd/libqpx_hip_nola.so qpth_amd/libqpx_hip_late.so >> $OUT/summary.txt 2>&1
echo "== A/B C5 shard (B=8192 n=64 m=64)" >> $OUT/summary.txt
timeout 600 python scripts/ab_bench.py qpth_amd/libqpx_hip_old.so qpth_amd/libqpx_hip.so 8192 64 64 0 >> $OUT/summary.txt 2>&1
echo "== A/B C3 (B=512 n=100 m=50 q=10)" >> $OUT/summary.txt
timeout 600 python scripts/ab_bench.py qpth_amd/libqpx_hip_old.so qpth_amd/libqpx_hip.so 512 100 50 10 >> $OUT/summary.txt 2>&1
echo "== phases (new build)" >> $OUT/summary.txt
timeout 300 python scripts/prof_phases.py >> $OUT/summary.txt 2>&1
timeout 300 python scripts/prof_panel.py >> $OUT/summary.txt 2>&1
echo "== pytest" >> $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
grep -E "f32 rel err|passed|failed|FAILED|Error|error" $OUT/pytest_gpu.log | tail -30 >> $OUT/summary.txt
