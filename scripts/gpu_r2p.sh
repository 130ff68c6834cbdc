#!/bin/bash
# GPU visit: issue priorities (MFMA streams at 0, everything else at 3) -- quick parity, same-box A/B
TAG=${1:-r02p}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "golden" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== A/B C2 (noprio = every phase at the default issue priority)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_noprio.so qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
echo "== A/B C3 shape (n=100 m=50 q=10)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_noprio.so qpth_amd/libqpx_hip.so 512 100 50 10 >> $OUT/summary.txt 2>&1
echo "== A/B C5 shape (n=64 m=64), B=8192" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_noprio.so qpth_amd/libqpx_hip.so 8192 64 64 0 >> $OUT/summary.txt 2>&1
