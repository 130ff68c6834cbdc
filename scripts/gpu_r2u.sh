#!/bin/bash
# Final GPU visit of round 2 (6.9 GPU-minutes left).  Build under test: QPX_F32_WIDE inside the kernels (ABI v4).
# 1. parity subset (everything but the full-size oracle runs, which r02t ran green on the same f64 kernels)
# 2. same-box A/B: HEAD's library (libqpx_hip_head.so) vs this build; host-side casts vs in-kernel widening
# 3. bench f64 / f32, rocprofv3 stats + PMC of this build (new kernel-source digest)
# 4. the reference's two timing tables and the other configurations' bench lines
TAG=${1:-r02u}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
REPO=$(pwd)
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== pytest -m gpu, subset" | tee $OUT/summary.txt
timeout 200 python -m pytest tests -m gpu -q -x --timeout 120 -s -k "not full_size and not c5_shard and not large_qp" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
grep -a "f32 rel err" $OUT/pytest_gpu.log | sort -u >> $OUT/summary.txt
tail -4 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== A/B C2 f64: HEAD (7db54c5 kernels) vs this build ($(el))" | tee -a $OUT/summary.txt
timeout 60 python scripts/ab_bench.py qpth_amd/libqpx_hip_head.so qpth_amd/libqpx_hip.so 2>&1 | grep -v amdgpu.ids | tee $PROF/${TAG}_ab_c2.txt >> $OUT/summary.txt
echo "== A/B float32 tensors: host-side casts vs QPX_F32_WIDE ($(el))" | tee -a $OUT/summary.txt
timeout 60 python scripts/ab_wide.py 2>&1 | grep -v amdgpu.ids | tee $PROF/${TAG}_ab_wide.txt >> $OUT/summary.txt
echo "== bench f64 ($(el))" | tee -a $OUT/summary.txt
timeout 120 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== rocprofv3 kernel stats + pmc ($(el))" | tee -a $OUT/summary.txt
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- $CMD > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $?" | tee -a $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline"; echo "# bench line of that run:"; grep '^{' $OUT/prof_stats.log | sed 's/^/# /';
  find /tmp/prof_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_kernel_stats.txt 2>&1
head -12 $PROF/${TAG}_kernel_stats.txt | cut -c1-200 >> $OUT/summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 100 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $?" | tee -a $OUT/summary.txt
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_pmc_$C.txt 2>&1
done
python scripts/make_traffic_json.py $PROF/${TAG}_pmc_FETCH_SIZE.txt $PROF/${TAG}_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json 2>> $OUT/summary.txt
cat $PROF/ipm_traffic.json >> $OUT/summary.txt
echo "== bench float32 tensors (default = QPX_F32_WIDE) ($(el))" | tee -a $OUT/summary.txt
timeout 100 python bench.py --dtype f32 --no-cpu-baseline > $OUT/bench_f32.json 2>> $OUT/bench.err
cat $OUT/bench_f32.json >> $OUT/summary.txt
echo "== tables ($(el))" | tee -a $OUT/summary.txt
timeout 120 python bench.py --table prof-linear 2>> $OUT/bench.err | tee $PROF/${TAG}_table_prof_linear.jsonl >> $OUT/summary.txt
timeout 60 python bench.py --table prof-gurobi 2>> $OUT/bench.err | tee $PROF/${TAG}_table_prof_gurobi.jsonl >> $OUT/summary.txt
echo "== other configurations ($(el))" | tee -a $OUT/summary.txt
timeout 60 python bench.py --config c3 --steps 100 --no-cpu-baseline 2>> $OUT/bench.err | tee $PROF/${TAG}_bench_c3.json >> $OUT/summary.txt
timeout 60 python bench.py --shared --steps 100 --no-cpu-baseline 2>> $OUT/bench.err | tee $PROF/${TAG}_bench_c2_shared.json >> $OUT/summary.txt
timeout 90 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tee $PROF/${TAG}_bench_c4.json >> $OUT/summary.txt
timeout 120 python bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline 2>> $OUT/bench.err | tee $PROF/${TAG}_bench_c5_one_gpu.json >> $OUT/summary.txt
echo "== done ($(el))" | tee -a $OUT/summary.txt
