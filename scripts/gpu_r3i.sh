#!/bin/bash
# Round 3: the tile sweep (pre-factorisation on matrix-core tiles).  Parity subset, then same-box A/B against the
# thread-grid sweep (variant + 32768) and the round-2 library.
TAG=${1:-r03i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== pytest -m gpu, subset ($(el))" | tee $OUT/summary.txt
timeout 300 python -m pytest tests -m gpu -q -x --timeout 200 -k "golden_batches or solver_entry or (against_oracle) or edge_shapes or reference_gradient" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log >> $OUT/summary.txt
for dims in "512 100 100 0" "512 100 50 10" "8192 64 64 0"; do
  echo "== A/B B n m q = $dims: round 2 | this build | this build with the thread-grid sweep ($(el))" >> $OUT/summary.txt
  timeout 200 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so qpth_amd/libqpx_hip.so:32768 $dims 2>&1 | grep -v amdgpu.ids | tail -6 >> $OUT/summary.txt
done
echo "== bench ($(el))" | tee -a $OUT/summary.txt
timeout 200 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== done ($(el))" | tee -a $OUT/summary.txt
