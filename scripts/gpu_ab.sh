#!/bin/bash
# Same-box A/B of library builds: gpu_ab.sh <tag> "<lib[:variant] ...>" ["B n m q" ...]
TAG=$1; LIBS=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
if [ $# -eq 0 ]; then set -- "512 100 100 0"; fi
for dims in "$@"; do
  echo "== B n m q = $dims" >> $OUT/summary.txt
  timeout 200 python scripts/ab_bench.py $LIBS $dims 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
