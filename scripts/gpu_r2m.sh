#!/bin/bash
# GPU visit: update phase without tile copies, pivot check after the block -- quick parity, same-box A/B, panel phases
TAG=${1:-r02m}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "golden or every_loop or hard" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== A/B C2 (p4 = four-column panels, bperm = group broadcast by ds_bpermute)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_bperm.so qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
echo "== panel phases" >> $OUT/summary.txt
timeout 300 python scripts/prof_panel.py >> $OUT/summary.txt 2>&1
echo "== loop phases" >> $OUT/summary.txt
[ -f qpth_amd/libqpx_hip_prof.so ] && timeout 300 python scripts/prof_phases.py 512 100 100 0 >> $OUT/summary.txt 2>&1
