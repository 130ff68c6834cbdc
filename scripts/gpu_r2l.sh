#!/bin/bash
# GPU visit: pivot block with four lane groups (lane swaps) -- parity subset with both group-broadcast forms, same-box A/B
TAG=${1:-r02l}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
K="golden or every_loop or reference_gradient or full_size_matches_oracle_c2 or edge"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "$K" > $OUT/pytest_gpu.log 2>&1; echo "pytest (lane swaps) exit $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log >> $OUT/summary.txt
cp qpth_amd/libqpx_hip.so /tmp/libqpx_hip_swap.so
cp qpth_amd/libqpx_hip_bperm.so qpth_amd/libqpx_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "$K" > $OUT/pytest_gpu_bperm.log 2>&1; echo "pytest (bpermute) exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_gpu_bperm.log >> $OUT/summary.txt
cp /tmp/libqpx_hip_swap.so qpth_amd/libqpx_hip.so
echo "== A/B C2 (p4 = four-column panels, bperm = group broadcast by ds_bpermute)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_bperm.so qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
echo "== A/B C3 shape (n=100 m=50 q=10)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so 512 100 50 10 >> $OUT/summary.txt 2>&1
echo "== panel phases" >> $OUT/summary.txt
timeout 300 python scripts/prof_panel.py >> $OUT/summary.txt 2>&1
