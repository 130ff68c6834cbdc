#!/bin/bash
# Round 4, visit c: the whole -m gpu suite on the round's sources, the bench lines of every configuration, rocprofv3
# kernel stats + the two HBM PMC passes of the default bench (-> ipm_traffic.json), C4 kernel stats + a FETCH_SIZE pass,
# the reference's tables, and a same-box A/B against the round-3 library (qpth_amd/libqpx_hip_r03.so).
# Summaries under gpurun_out/$TAG/profiles are what gets copied to profiles/.
TAG=${1:-r04c}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
REPO=$(pwd)
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $? ($(el))" | tee -a $OUT/summary.txt
tail -2 $OUT/smoke.log >> $OUT/summary.txt
echo "== bench (default: C2 f64)" | tee -a $OUT/summary.txt
timeout 300 python bench.py > $PROF/${TAG}_bench_f64.json 2> $OUT/bench.err; echo "bench exit $? ($(el))" | tee -a $OUT/summary.txt
cat $PROF/${TAG}_bench_f64.json >> $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -s --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
grep -a "rel err\|float32, rel err" $OUT/pytest_gpu.log | sort -u >> $OUT/summary.txt
tail -25 $OUT/pytest_gpu.log >> $OUT/summary.txt
cp $OUT/pytest_gpu.log $PROF/${TAG}_pytest_gpu.txt
if [ -z "$SKIP_PROF" ]; then
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- $CMD > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $? ($(el))" | tee -a $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline"; echo "# bench line of that run:"; grep '^{' $OUT/prof_stats.log | sed 's/^/# /';
  find /tmp/prof_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_kernel_stats.txt 2>&1
head -30 $PROF/${TAG}_kernel_stats.txt | cut -c1-300 >> $OUT/summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $? ($(el))" | tee -a $OUT/summary.txt
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_pmc_$C.txt 2>&1
done
python scripts/make_traffic_json.py $PROF/${TAG}_pmc_FETCH_SIZE.txt $PROF/${TAG}_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json 2>> $OUT/summary.txt
cat $PROF/ipm_traffic.json >> $OUT/summary.txt
fi
echo "== other configurations" | tee -a $OUT/summary.txt
timeout 200 python bench.py --dtype f32 --no-cpu-baseline > $PROF/${TAG}_bench_f32.json 2>> $OUT/bench.err
timeout 200 python bench.py --dtype f32 --refine 0 --no-cpu-baseline > $PROF/${TAG}_bench_f32_refine0.json 2>> $OUT/bench.err
timeout 200 python bench.py --dtype f32 --refine 2 --no-cpu-baseline > $PROF/${TAG}_bench_f32_refine2.json 2>> $OUT/bench.err
timeout 200 python bench.py --shared --no-cpu-baseline > $PROF/${TAG}_bench_shared.json 2>> $OUT/bench.err
timeout 200 python bench.py --config c3 --no-cpu-baseline > $PROF/${TAG}_bench_c3.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_c4.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c4 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_c4_f32.json 2>> $OUT/bench.err
timeout 300 python bench.py --config custom --batch 128 --nz 500 --nineq 400 --neq 100 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_c4_neq100.json 2>> $OUT/bench.err
timeout 300 python bench.py --config custom --batch 512 --nz 150 --nineq 150 --neq 0 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_b512_n150_m150.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $PROF/${TAG}_bench_c5_one_gpu.json 2>> $OUT/bench.err
for f in f32 f32_refine0 f32_refine2 shared c3 c4 c4_f32 c4_neq100 b512_n150_m150 c5_one_gpu; do echo "-- $f" >> $OUT/summary.txt; cut -c1-700 $PROF/${TAG}_bench_$f.json >> $OUT/summary.txt; grep -o '"kernel_ms": {[^}]*}' $PROF/${TAG}_bench_$f.json >> $OUT/summary.txt; done
echo "($(el))" | tee -a $OUT/summary.txt
if [ -z "$SKIP_PROF" ]; then
echo "== C4: rocprofv3 kernel stats + FETCH_SIZE" | tee -a $OUT/summary.txt
CMD4="python $REPO/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats4 -o stats -- $CMD4 > $REPO/$OUT/prof_stats4.log 2>&1); echo "rocprof c4 stats exit $? ($(el))" | tee -a $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"; grep '^{' $OUT/prof_stats4.log | cut -c1-600 | sed 's/^/# /';
  find /tmp/prof_stats4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; python scripts/rocprof_timeline.py "$f" --last 800 > $PROF/${TAG}_c4_timeline.txt 2>&1; done; } > $PROF/${TAG}_c4_kernel_stats.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_pmc4 -o pmc -- $CMD4 > $REPO/$OUT/prof_pmc4.log 2>&1); echo "pmc c4 exit $? ($(el))" | tee -a $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
  find /tmp/prof_pmc4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_c4_pmc_FETCH_SIZE.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_pmc4w -o pmc -- $CMD4 > $REPO/$OUT/prof_pmc4w.log 2>&1); echo "pmc c4 write exit $? ($(el))" | tee -a $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
  find /tmp/prof_pmc4w -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_c4_pmc_WRITE_SIZE.txt 2>&1
head -24 $PROF/${TAG}_c4_kernel_stats.txt | cut -c1-300 >> $OUT/summary.txt
head -14 $PROF/${TAG}_c4_pmc_FETCH_SIZE.txt | cut -c1-300 >> $OUT/summary.txt
echo "== the reference's tables" | tee -a $OUT/summary.txt
timeout 300 python bench.py --table prof-linear > $PROF/${TAG}_table_prof_linear.jsonl 2>> $OUT/bench.err
timeout 300 python bench.py --table prof-gurobi > $PROF/${TAG}_table_prof_gurobi.jsonl 2>> $OUT/bench.err
echo "tables: $(wc -l < $PROF/${TAG}_table_prof_linear.jsonl) + $(wc -l < $PROF/${TAG}_table_prof_gurobi.jsonl) rows ($(el))" | tee -a $OUT/summary.txt
fi
if [ -f qpth_amd/libqpx_hip_r03.so ]; then
echo "== A/B on this box against the round-3 build" | tee -a $OUT/summary.txt
for dims in "512 100 100 0" "512 100 50 10" "128 500 500 0" "8192 64 64 0"; do
  echo "-- B n m q = $dims" >> $PROF/${TAG}_ab_r03.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r03.so qpth_amd/libqpx_hip.so $dims 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_ab_r03.txt
done
cat $PROF/${TAG}_ab_r03.txt >> $OUT/summary.txt
fi
tail -5 $OUT/bench.err >> $OUT/summary.txt
echo "== finishing kernel: time by (steps, refine)" | tee -a $OUT/summary.txt
timeout 200 python scripts/prof_polish.py 2>&1 | grep -v amdgpu.ids | tee $PROF/${TAG}_polish_steps.txt >> $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
echo "== pre-factorisation: matrix cores (default) against the sweep (knob bit 14), phases of the default" | tee -a $OUT/summary.txt
for dims in "512 100 100 0" "2048 100 100 0" "512 64 64 0" "65536 64 64 0"; do
  echo "-- B n m q = $dims" >> $PROF/${TAG}_ab_prefac.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:16384 qpth_amd/libqpx_hip.so:0 $dims 2>&1 | grep -v amdgpu.ids | tail -4 >> $PROF/${TAG}_ab_prefac.txt
done
[ -f qpth_amd/libqpx_hip_prof.so ] && timeout 200 python scripts/prof_prefac.py 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_ab_prefac.txt
cat $PROF/${TAG}_ab_prefac.txt >> $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
