#!/bin/bash
# Round 4, visit w: the whole -m gpu suite on the round's final sources (after r04v's change), nothing else
TAG=${1:-r04w}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 560 python -m pytest tests -m gpu -q --timeout 300 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" > $OUT/summary.txt
tail -6 $OUT/pytest_gpu.log >> $OUT/summary.txt
cat $OUT/summary.txt
