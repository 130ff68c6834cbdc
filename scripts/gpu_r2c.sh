#!/bin/bash
# Round-2 third GPU visit: the large-QP family (C4) -- parity, bench, per-kernel breakdown.
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -s -k "c4 or large_qp or loop_kernel_form or needs_input" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
grep -E "passed|failed|FAILED|Error|error|assert" $OUT/pytest_gpu.log | tail -20 >> $OUT/summary.txt
echo "== bench c4" >> $OUT/summary.txt
timeout 600 python bench.py --config c4 --steps 20 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench c4 exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench_c4.json >> $OUT/summary.txt; tail -3 $OUT/bench_c4.err >> $OUT/summary.txt
echo "== bench c2 (sanity)" >> $OUT/summary.txt
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/summary.txt 2>&1
echo "== rocprofv3 kernel stats, bench c4" >> $OUT/summary.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o stats -- python $REPO/bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_c4.log 2>&1); echo "rocprof exit $?" | tee -a $OUT/summary.txt
find /tmp/prof_c4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done > $OUT/c4_kernel_stats.txt 2>&1
cat $OUT/c4_kernel_stats.txt >> $OUT/summary.txt
