#!/bin/bash
# GPU visit: padded pivots skipped + pivot block read back before the publish; two waves per QP at C2 (no spills any more)
TAG=${1:-r02s}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "golden or every_loop or hard or edge" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== A/B C2 (r02q = previous build)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02q.so qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
echo "== C2, two waves per QP (QPX_VARIANT=5120), one wave (3072)" >> $OUT/summary.txt
QPX_VARIANT=5120 timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
QPX_VARIANT=3072 timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
echo "== A/B C3 shape (n=100 m=50 q=10): default, two waves, four waves" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02q.so qpth_amd/libqpx_hip.so 512 100 50 10 >> $OUT/summary.txt 2>&1
QPX_VARIANT=5120 timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so 512 100 50 10 >> $OUT/summary.txt 2>&1
QPX_VARIANT=9216 timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so 512 100 50 10 >> $OUT/summary.txt 2>&1
echo "== small shapes: B=4096 n=10 m=5; n=m=32" >> $OUT/summary.txt
timeout 200 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_r02q.so qpth_amd/libqpx_hip.so 4096 10 5 0 >> $OUT/summary.txt 2>&1
timeout 200 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so 4096 32 32 0 >> $OUT/summary.txt 2>&1
echo "== C5 shape (n=m=64, B=8192): default, two waves" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02q.so qpth_amd/libqpx_hip.so 8192 64 64 0 >> $OUT/summary.txt 2>&1
QPX_VARIANT=5120 timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so 8192 64 64 0 >> $OUT/summary.txt 2>&1
