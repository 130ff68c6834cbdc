#!/bin/bash
# Round 5, visit d: large-QP family with the symmetric mat-vec (R read by its lower triangle only, no mirrored R) -- parts
# WITHOUT helper streams against one part with its helper, four shapes, same box; the family's parity tests; C4 bench + stats.
TAG=${1:-r05d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
S=$OUT/summary.txt
echo "== parts (65536 = one part + helper stream, 131072 / 196608 / 262144 = 2 / 3 / 4 parts, 0 = automatic); r04 = round-4 library" > $S
for dims in "128 500 500 0" "512 150 150 0" "32 500 500 0" "64 300 300 0" "16 300 300 20" "256 200 200 0"; do
  echo "-- B n m q = $dims" >> $S
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r04.so:0 qpth_amd/libqpx_hip.so:65536 qpth_amd/libqpx_hip.so:131072 qpth_amd/libqpx_hip.so:196608 qpth_amd/libqpx_hip.so:262144 qpth_amd/libqpx_hip.so:0 $dims 2>&1 | grep -v amdgpu.ids | tail -12 >> $S
done
cp $S $OUT/ab_parts.txt
echo "== pytest (large-QP family)" >> $S
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "full_size_matches_oracle_c4 or large_qps_with_equality or accuracy_options or c4_float32 or solver_entry_points or every_loop_kernel_form or regularised" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -6 >> $S
echo "== bench c4" >> $S
timeout 600 python bench.py --config c4 --steps 20 --warmup 3 > $OUT/bench_c4.json 2> $OUT/bench.err; echo "bench c4 exit $?" >> $S; cut -c1-400 $OUT/bench_c4.json >> $S
CMD="python $REPO/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o stats -- $CMD > $REPO/$OUT/prof_c4.log 2>&1); echo "rocprof exit $?" >> $S
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"; grep '^{' $OUT/prof_c4.log | sed 's/^/# /' | cut -c1-600;
  find /tmp/prof_c4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $OUT/${TAG}_c4_kernel_stats.txt 2>&1
cat $OUT/${TAG}_c4_kernel_stats.txt >> $S
