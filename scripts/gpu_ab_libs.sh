#!/bin/bash
# same-box A/B of several builds of the library: gpu_ab_libs.sh TAG "lib1 lib2 ..." ["B n m q" ...]
TAG=$1; LIBS=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "512 100 100 0" "512 100 50 10" "8192 64 64 0"; fi
for dims in "$@"; do
  echo "-- B n m q = $dims" >> $OUT/${TAG}_ab.txt
  timeout 400 python scripts/ab_bench.py $LIBS $dims 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_ab.txt
done
cat $OUT/${TAG}_ab.txt
