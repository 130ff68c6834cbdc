#!/bin/bash
# Same-box A/B of library builds, ONE PROCESS PER LIBRARY (alternating, three rounds): needed when both builds fork side
# streams (the large-QP family): two libraries in one process hold two stream pools, and the streams of the second share
# hardware queues with the first's (profiles/r05f_*.txt).    ab_procs.sh "<lib[:variant] ...>" "B n m q" ...
LIBS=$1; shift
for dims in "$@"; do
  echo "-- B n m q = $dims"
  for rep in 1 2 3; do
    for lib in $LIBS; do
      timeout 200 python scripts/ab_bench.py $lib $dims 2>&1 | grep -v amdgpu.ids | tail -1
    done
  done
done
