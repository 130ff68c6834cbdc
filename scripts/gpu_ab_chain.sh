#!/bin/bash
# Round 5: the chain form of a pair of panels (three launches, default) against the four-launch order (knob bit 25) and the
# round-4 library on the same box; the family's parity tests; C4 bench + kernel stats.
TAG=${1:-r05e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
S=$OUT/summary.txt
echo "== chain form (0) vs four launches per pair (33554432 = bit 25) vs round-4 library" > $S
for dims in "128 500 500 0" "512 150 150 0" "128 500 400 100" "64 300 300 0" "256 200 200 0"; do
  echo "-- B n m q = $dims" >> $S
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r04.so:0 qpth_amd/libqpx_hip.so:33554432 qpth_amd/libqpx_hip.so:0 $dims 2>&1 | grep -v amdgpu.ids | tail -9 >> $S
done
cp $S $OUT/ab_chain.txt
echo "== pytest (large-QP family)" >> $S
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "full_size_matches_oracle_c4 or large_qps_with_equality or accuracy_options or c4_float32 or solver_entry_points or every_loop_kernel_form or regularised or refinement_is_refused" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -6 >> $S
echo "== bench c4" >> $S
timeout 600 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench.err; echo "bench c4 exit $?" >> $S; cut -c1-400 $OUT/bench_c4.json >> $S
grep -o '"kernel_ms": {[^}]*}' $OUT/bench_c4.json >> $S
CMD="python $REPO/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o stats -- $CMD > $REPO/$OUT/prof_c4.log 2>&1); echo "rocprof exit $?" >> $S
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"; grep '^{' $OUT/prof_c4.log | sed 's/^/# /' | cut -c1-600;
  find /tmp/prof_c4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; python scripts/rocprof_timeline.py "$f" --last 800 > $OUT/${TAG}_c4_timeline.txt 2>&1; done; } > $OUT/${TAG}_c4_kernel_stats.txt 2>&1
cat $OUT/${TAG}_c4_kernel_stats.txt >> $S
head -30 $OUT/${TAG}_c4_timeline.txt >> $S
