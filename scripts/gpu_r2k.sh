#!/bin/bash
# GPU visit: sixteen-column panels of the tile kernels -- parity subset, same-box A/B against the four-column build
# (libqpx_hip_p4.so = k9 with -DQPX_TILE_PANEL4), sub-phases of a panel, bench line
TAG=${1:-r02k}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "${KEXPR:-reference_gradient or golden or entry_points or against_oracle or full_size or every_loop or hard or edge or c5}" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== A/B (four-column panels = p4)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
echo "== A/B C3 shape (n=100 m=50 q=10)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so 512 100 50 10 >> $OUT/summary.txt 2>&1
echo "== A/B C5 shape (n=64 m=64), B=8192" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so 8192 64 64 0 >> $OUT/summary.txt 2>&1
echo "== panel phases" >> $OUT/summary.txt
timeout 300 python scripts/prof_panel.py >> $OUT/summary.txt 2>&1
echo "== bench" >> $OUT/summary.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json >> $OUT/summary.txt; tail -3 $OUT/bench.err >> $OUT/summary.txt
