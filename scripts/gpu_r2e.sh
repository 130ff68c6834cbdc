#!/bin/bash
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -s -k "c4 or large_qp or loop_kernel_form or float32 or refinement" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
grep -E "f32 rel err|KKT residual|passed|failed|FAILED|Error|error" $OUT/pytest_gpu.log | tail -20 >> $OUT/summary.txt
echo "== bench c4" >> $OUT/summary.txt
timeout 600 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline >> $OUT/summary.txt 2>&1
echo "== bench c2 f32 (default refine)" >> $OUT/summary.txt
timeout 300 python bench.py --dtype f32 --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/summary.txt 2>&1
echo "== rocprofv3 kernel stats, bench c4" >> $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o stats -- python $REPO/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_c4.log 2>&1); echo "rocprof exit $?" | tee -a $OUT/summary.txt
find /tmp/prof_c4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done > $OUT/c4_kernel_stats.txt 2>&1
cat $OUT/c4_kernel_stats.txt >> $OUT/summary.txt
