#!/bin/bash
# Round 4, visit x: the rest of the -m gpu suite (from the test visit w stopped at) on the round's final sources
TAG=${1:-r04x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 310 python -m pytest $(cat scripts/.r04x_ids.txt | tr '\n' ' ') -q --timeout 200 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" > $OUT/summary.txt
tail -6 $OUT/pytest_gpu.log >> $OUT/summary.txt
cat $OUT/summary.txt
