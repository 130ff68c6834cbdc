#!/bin/bash
# Round 6 (from scripts/gpu_r5_final.sh), the full visit on the round's final sources: smoke, the whole -m gpu suite, the bench lines of every
# configuration, rocprofv3 kernel stats + the two HBM PMC passes of the default bench (-> ipm_traffic.json) and of C4, the
# reference's tables, same-box A/Bs against the round-4 library (qpth_amd/libqpx_hip_r05.so) and of the parts knob.
# Summaries under gpurun_out/$TAG/profiles are what gets copied to profiles/.
TAG=${1:-r06z}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
REPO=$(pwd)
S=$OUT/summary.txt
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== smoke" > $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $? ($(el))" >> $S
tail -2 $OUT/smoke.log >> $S
echo "== bench (default: C2 f64; extra: C3, C4)" >> $S
timeout 400 python bench.py > $PROF/${TAG}_bench_f64.json 2> $OUT/bench.err; echo "bench exit $? ($(el))" >> $S
cat $PROF/${TAG}_bench_f64.json >> $S
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu" >> $S
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -s --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(el))" >> $S
grep -a "rel err\|float32 kernels vs\|float32 tensors vs" $OUT/pytest_gpu.log | sort -u >> $S
grep -v amdgpu.ids $OUT/pytest_gpu.log | tail -18 >> $S
grep -v amdgpu.ids $OUT/pytest_gpu.log > $PROF/${TAG}_pytest_gpu.txt
fi
echo "== rocprofv3 kernel stats + PMC (default bench)" >> $S
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --step-kernels-only"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- $CMD > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $? ($(el))" >> $S
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --step-kernels-only"; echo "# bench line of that run:"; grep '^{' $OUT/prof_stats.log | sed 's/^/# /';
  find /tmp/prof_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_kernel_stats.txt 2>&1
head -24 $PROF/${TAG}_kernel_stats.txt | cut -c1-300 >> $S
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs --step-kernels-only > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $? ($(el))" >> $S
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs --step-kernels-only"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_pmc_$C.txt 2>&1
done
python scripts/make_traffic_json.py $PROF/${TAG}_pmc_FETCH_SIZE.txt $PROF/${TAG}_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json 2>> $S
cat $PROF/ipm_traffic.json >> $S
cp gpurun_out/nccl_world1_bench.json $PROF/${TAG}_nccl_world1_bench.json 2>/dev/null
echo "== other configurations" >> $S
timeout 200 python bench.py --dtype f32 --no-cpu-baseline > $PROF/${TAG}_bench_f32.json 2>> $OUT/bench.err
timeout 200 python bench.py --dtype f32 --refine 0 --no-cpu-baseline > $PROF/${TAG}_bench_f32_refine0.json 2>> $OUT/bench.err
timeout 200 python bench.py --dtype f32 --refine 2 --no-cpu-baseline > $PROF/${TAG}_bench_f32_refine2.json 2>> $OUT/bench.err
timeout 200 python bench.py --shared --no-cpu-baseline > $PROF/${TAG}_bench_shared.json 2>> $OUT/bench.err
timeout 200 python bench.py --config c3 --no-cpu-baseline > $PROF/${TAG}_bench_c3.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c4 --steps 20 --warmup 3 > $PROF/${TAG}_bench_c4.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c4 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_c4_f32.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c4 --dtype f32 --refine 2 --steps 10 --warmup 2 --no-cpu-baseline > $PROF/${TAG}_bench_c4_f32_refine2.json 2>> $OUT/bench.err
timeout 300 python bench.py --config custom --batch 128 --nz 500 --nineq 400 --neq 100 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_c4_neq100.json 2>> $OUT/bench.err
timeout 300 python bench.py --config custom --batch 512 --nz 150 --nineq 150 --neq 0 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_b512_n150_m150.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $PROF/${TAG}_bench_c5_one_gpu.json 2>> $OUT/bench.err
for f in f32 f32_refine0 f32_refine2 shared c3 c4 c4_f32 c4_f32_refine2 c4_neq100 b512_n150_m150 c5_one_gpu; do echo "-- $f" >> $S; cut -c1-420 $PROF/${TAG}_bench_$f.json >> $S; grep -o '"kernel_ms": {[^}]*}' $PROF/${TAG}_bench_$f.json >> $S; done
echo "($(el))" >> $S
echo "== C4: rocprofv3 kernel stats + PMC" >> $S
CMD4="python $REPO/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats4 -o stats -- $CMD4 > $REPO/$OUT/prof_stats4.log 2>&1); echo "rocprof c4 stats exit $? ($(el))" >> $S
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"; grep '^{' $OUT/prof_stats4.log | cut -c1-600 | sed 's/^/# /';
  find /tmp/prof_stats4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; python scripts/rocprof_timeline.py "$f" --last 800 > $PROF/${TAG}_c4_timeline.txt 2>&1; done; } > $PROF/${TAG}_c4_kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc4_$C -o pmc -- $CMD4 > $REPO/$OUT/prof_pmc4_$C.log 2>&1); echo "pmc c4 $C exit $? ($(el))" >> $S
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc4_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_c4_pmc_$C.txt 2>&1
done
head -22 $PROF/${TAG}_c4_kernel_stats.txt | cut -c1-300 >> $S
echo "== the reference's tables" >> $S
timeout 300 python bench.py --table prof-linear > $PROF/${TAG}_table_prof_linear.jsonl 2>> $OUT/bench.err
timeout 300 python bench.py --table prof-gurobi > $PROF/${TAG}_table_prof_gurobi.jsonl 2>> $OUT/bench.err
echo "tables: $(wc -l < $PROF/${TAG}_table_prof_linear.jsonl) + $(wc -l < $PROF/${TAG}_table_prof_gurobi.jsonl) rows ($(el))" >> $S
if [ -f qpth_amd/libqpx_hip_r05.so ]; then
echo "== A/B on this box against the round-5 build" >> $S
for dims in "512 100 100 0" "512 100 50 10" "128 500 500 0" "512 150 150 0" "8192 64 64 0" "65536 64 64 0"; do
  echo "-- B n m q = $dims" >> $PROF/${TAG}_ab_r05.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r05.so qpth_amd/libqpx_hip.so $dims 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_ab_r05.txt
done
cat $PROF/${TAG}_ab_r05.txt >> $S
fi
echo "== large-QP family: one part (knob bits 16..19 = 1 -> :65536) against the default two parts, by entry point (step, loop, pre-factorisation; backward = the rest)" >> $S
for dims in "128 500 500 0" "512 150 150 0" "128 300 200 50"; do
  echo "-- B n m q = $dims" >> $PROF/${TAG}_ab_parts.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:65536 qpth_amd/libqpx_hip.so:0 $dims 2>&1 | grep -v amdgpu.ids | tail -4 >> $PROF/${TAG}_ab_parts.txt
done
cat $PROF/${TAG}_ab_parts.txt >> $S
echo "== finishing stage: time by (steps, refine)" >> $S
{ timeout 200 python scripts/prof_polish.py; timeout 300 python scripts/prof_polish.py 128 500 500 0; } 2>&1 | grep -v amdgpu.ids | tee $PROF/${TAG}_polish_steps.txt >> $S
tail -5 $OUT/bench.err >> $S
du -sh $OUT >> $S
