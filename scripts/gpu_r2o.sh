#!/bin/bash
# GPU visit: what the 16 x 16 pivot block costs alone / beside an MFMA-streaming wave; A/B of s_setprio around it and of
# one Newton step less in the reciprocal (timing only)
TAG=${1:-r02o}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/ubench_pivot.py > $OUT/ubench_pivot.txt 2>&1; cat $OUT/ubench_pivot.txt | tee $OUT/summary.txt
echo "== A/B C2 (prio = s_setprio 3 around the pivot block, newton1 = one Newton step in rcp_)" >> $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_prio.so qpth_amd/libqpx_hip_newton1.so qpth_amd/libqpx_hip.so >> $OUT/summary.txt 2>&1
