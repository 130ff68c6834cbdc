#!/bin/bash
# Last GPU visit of round 2 (13 GPU-minutes left): the checks of scripts/gpu_check.sh in order of importance, each
# bounded, so that a call cut short by the budget still leaves the parity run and the f64 bench line behind.
# New in this build: float32 tensors in float64 arithmetic (QPFunction(refine=None) at the tile-kernel sizes).
TAG=${1:-r02t}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
REPO=$(pwd)
date +%s > $OUT/t0
echo "== smoke" | tee $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt
tail -2 $OUT/smoke.log >> $OUT/summary.txt
echo "== pytest -m gpu ($(( $(date +%s) - $(cat $OUT/t0) )) s)" | tee -a $OUT/summary.txt
timeout 420 python -m pytest tests -m gpu -q -x --timeout 300 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
grep -a "f32 rel err" $OUT/pytest_gpu.log >> $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== bench f64 ($(( $(date +%s) - $(cat $OUT/t0) )) s)" | tee -a $OUT/summary.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== rocprofv3 kernel stats ($(( $(date +%s) - $(cat $OUT/t0) )) s)" | tee -a $OUT/summary.txt
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- $CMD > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $?" | tee -a $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline"; echo "# bench line of that run:"; grep '^{' $OUT/prof_stats.log | sed 's/^/# /';
  find /tmp/prof_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_kernel_stats.txt 2>&1
head -30 $PROF/${TAG}_kernel_stats.txt | cut -c1-220 >> $OUT/summary.txt
echo "== rocprofv3 pmc ($(( $(date +%s) - $(cat $OUT/t0) )) s)" | tee -a $OUT/summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $?" | tee -a $OUT/summary.txt
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_pmc_$C.txt 2>&1
done
python scripts/make_traffic_json.py $PROF/${TAG}_pmc_FETCH_SIZE.txt $PROF/${TAG}_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json 2>> $OUT/summary.txt
cat $PROF/ipm_traffic.json >> $OUT/summary.txt
echo "== bench f32 tensors: default (f64 arithmetic), refine=0 (f32 kernels alone), refine=2 ($(( $(date +%s) - $(cat $OUT/t0) )) s)" | tee -a $OUT/summary.txt
timeout 200 python bench.py --dtype f32 --no-cpu-baseline > $OUT/bench_f32.json 2>> $OUT/bench.err
cat $OUT/bench_f32.json >> $OUT/summary.txt
timeout 200 python bench.py --dtype f32 --refine 0 --steps 100 --no-cpu-baseline > $OUT/bench_f32_refine0.json 2>> $OUT/bench.err
cat $OUT/bench_f32_refine0.json >> $OUT/summary.txt
timeout 200 python bench.py --dtype f32 --refine 2 --steps 100 --no-cpu-baseline > $OUT/bench_f32_refine2.json 2>> $OUT/bench.err
cat $OUT/bench_f32_refine2.json >> $OUT/summary.txt
echo "== done ($(( $(date +%s) - $(cat $OUT/t0) )) s)" | tee -a $OUT/summary.txt
