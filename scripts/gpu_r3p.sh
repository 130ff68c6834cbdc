#!/bin/bash
# Round 3: the whole -m gpu suite on the current build, then the bench line
TAG=${1:-r03p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -s --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(el))" | tee $OUT/summary.txt
grep -a "rel err\|float32, rel err" $OUT/pytest_gpu.log | sort -u >> $OUT/summary.txt
tail -30 $OUT/pytest_gpu.log >> $OUT/summary.txt
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? ($(el))" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt
