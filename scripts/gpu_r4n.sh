#!/bin/bash
# Round 4, visit n (short): phases of the matrix-core pre-factorisation + A/B against the sweep
TAG=${1:-r04n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
timeout 200 python scripts/prof_prefac.py 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
timeout 200 python scripts/prof_prefac.py 256 100 100 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
timeout 200 python scripts/prof_prefac.py 512 64 64 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
L=qpth_amd/libqpx_hip.so
for dims in "512 100 100 0" "256 100 100 0" "512 64 64 0" "8192 64 64 0"; do
  echo "== $dims" >> $OUT/summary.txt
  timeout 200 python scripts/ab_bench.py $L:0 $L:16384 $dims 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT/summary.txt
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "c2 or spd" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/summary.txt
tail -2 $OUT/pytest.log >> $OUT/summary.txt
cat $OUT/summary.txt
