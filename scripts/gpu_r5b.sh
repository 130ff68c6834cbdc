#!/bin/bash
# Round 5, visit b: large-QP family -- the batch in 2 / 3 / 4 concurrent parts (knob bits 16..19) against one part, same
# box; per-kernel HBM bytes of the C4 forward+backward (PMC FETCH_SIZE / WRITE_SIZE, separate passes); the two-stage batch
# contraction on the GPU (parity + time at one GPU's share of C5).
TAG=${1:-r05b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
S=$OUT/summary.txt
echo "== C4: parts" > $S
timeout 400 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:0 qpth_amd/libqpx_hip.so:131072 qpth_amd/libqpx_hip.so:196608 qpth_amd/libqpx_hip.so:262144 qpth_amd/libqpx_hip.so:$((131072 + 2097152)) 128 500 500 0 2>&1 | grep -v amdgpu.ids > $OUT/ab_c4_parts.txt; cat $OUT/ab_c4_parts.txt >> $S
echo "== pytest two-stage contraction" >> $S
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 200 -k "two_stages or shared" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -5 >> $S
echo "== bench --shared (C2 shapes, B = 512 and 8192)" >> $S
timeout 300 python bench.py --shared --no-cpu-baseline > $OUT/bench_shared.json 2> $OUT/bench.err; cut -c1-700 $OUT/bench_shared.json >> $S
timeout 300 python bench.py --shared --no-cpu-baseline --batch 8192 --nz 64 --nineq 64 --steps 30 --warmup 5 > $OUT/bench_shared_8192.json 2>> $OUT/bench.err; cut -c1-700 $OUT/bench_shared_8192.json >> $S
echo "== rocprofv3 pmc C4" >> $S
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $?" >> $S
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $OUT/${TAG}_c4_pmc_$C.txt 2>&1
  cat $OUT/${TAG}_c4_pmc_$C.txt >> $S
done
tail -3 $OUT/bench.err >> $S
