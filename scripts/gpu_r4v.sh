#!/bin/bash
# Round 4, visit v (reduced final visit): the pre-factorisation with equality constraints on the matrix cores -- A/B
# against the previous build (sweep at neq > 0) on the same box, the parity tests that run through it, the default bench
# line, C3's, rocprofv3 kernel stats and the two HBM PMC passes of the final build (-> ipm_traffic.json).
TAG=${1:-r04v}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
REPO=$(pwd)
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
: > $OUT/summary.txt
if [ -f qpth_amd/libqpx_hip_prev.so ]; then
for dims in "512 100 50 10" "512 100 100 0" "512 60 70 6" "2048 100 50 10"; do
  echo "-- B n m q = $dims" >> $PROF/${TAG}_ab_prev.txt
  timeout 200 python scripts/ab_bench.py qpth_amd/libqpx_hip_prev.so qpth_amd/libqpx_hip.so $dims 2>&1 | grep -v amdgpu.ids | tail -4 >> $PROF/${TAG}_ab_prev.txt
done
cat $PROF/${TAG}_ab_prev.txt >> $OUT/summary.txt
fi
echo "($(el))" >> $OUT/summary.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "matrix_core or spd or c3 or full_size_c2 or entry_points" > $OUT/pytest.log 2>&1; echo "pytest exit $? ($(el))" >> $OUT/summary.txt
tail -4 $OUT/pytest.log >> $OUT/summary.txt; cp $OUT/pytest.log $PROF/${TAG}_pytest_gpu_subset.txt
timeout 200 python bench.py --no-cpu-baseline > $PROF/${TAG}_bench_f64.json 2> $OUT/bench.err; cut -c1-260 $PROF/${TAG}_bench_f64.json >> $OUT/summary.txt
timeout 200 python bench.py --config c3 --no-cpu-baseline > $PROF/${TAG}_bench_c3.json 2>> $OUT/bench.err; cut -c1-260 $PROF/${TAG}_bench_c3.json >> $OUT/summary.txt
grep -o '"kernel_ms": {[^}]*}' $PROF/${TAG}_bench_f64.json $PROF/${TAG}_bench_c3.json >> $OUT/summary.txt
echo "($(el))" >> $OUT/summary.txt
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- $CMD > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $? ($(el))" >> $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline"; grep '^{' $OUT/prof_stats.log | cut -c1-300 | sed 's/^/# /';
  find /tmp/prof_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $? ($(el))" >> $OUT/summary.txt
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_pmc_$C.txt 2>&1
done
python scripts/make_traffic_json.py $PROF/${TAG}_pmc_FETCH_SIZE.txt $PROF/${TAG}_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json 2>> $OUT/summary.txt
head -8 $PROF/${TAG}_kernel_stats.txt | cut -c1-160 >> $OUT/summary.txt
cat $OUT/summary.txt
