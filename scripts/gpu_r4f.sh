#!/bin/bash
# Round 4, visit f (short): the finishing kernel after its column products moved to all threads -- time by (steps,
# refine), the float32 refine=2 bench line, and its parity tests.
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/summary.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -s -k "float32_error_distribution or every_kkt_solver or solve_kkt_ir" > $OUT/pytest.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
grep -a "rel err" $OUT/pytest.log | cut -c1-400 >> $OUT/summary.txt; tail -3 $OUT/pytest.log >> $OUT/summary.txt
timeout 200 python scripts/prof_polish.py 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
timeout 200 python scripts/prof_polish.py 512 100 50 10 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
timeout 200 python bench.py --dtype f32 --refine 2 --no-cpu-baseline > $OUT/bench_f32_refine2.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench_f32_refine2.json >> $OUT/summary.txt; grep -o '"kernel_ms": {[^}]*}' $OUT/bench_f32_refine2.json >> $OUT/summary.txt
