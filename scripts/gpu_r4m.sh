#!/bin/bash
# Round 4, visit m (short): pre_factor_kkt on the matrix cores (qpx_prefac.h) against the thread-grid sweep (knob bit 14)
# on the same box -- step, loop kernel, pre-factorisation -- and the parity tests that run through it.
TAG=${1:-r04m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "c2 or spd or float32 or every_loop or shared or full_size_c5 or c5" > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -4 $OUT/pytest.log >> $OUT/summary.txt
L=qpth_amd/libqpx_hip.so
for dims in "512 100 100 0" "2048 100 100 0" "512 100 50 0" "512 64 64 0" "8192 64 64 0" "512 112 112 0" "512 50 100 0"; do
  echo "== $dims" >> $OUT/summary.txt
  timeout 200 python scripts/ab_bench.py $L:0 $L:16384 $dims 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT/summary.txt
done
cat $OUT/summary.txt
