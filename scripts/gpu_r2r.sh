#!/bin/bash
# GPU visit: the large-QP family (its 64 x 64 diagonal blocks run TileMat<4,1>::ldl_inv) with four- and sixteen-column panels
TAG=${1:-r02r}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== A/B C4 (B=128 nz=nineq=500), p4 = four-column panels" > $OUT/summary.txt
timeout 400 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so 128 500 500 0 >> $OUT/summary.txt 2>&1
echo "== bench --config c4" >> $OUT/summary.txt
timeout 400 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2>> $OUT/summary.txt; cat $OUT/bench_c4.json >> $OUT/summary.txt
echo "== A/B small shapes: C1-like B=4096 n=10 m=5; n=m=32 B=4096" >> $OUT/summary.txt
timeout 200 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so 4096 10 5 0 >> $OUT/summary.txt 2>&1
timeout 200 python scripts/ab_bench.py qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip.so 4096 32 32 0 >> $OUT/summary.txt 2>&1
