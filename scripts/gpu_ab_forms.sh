#!/bin/bash
# A/B of the loop-kernel forms: QPX_VARIANT 0 = auto, 256 = 16x16 thread grid, 512 = 8x8, 1024 = matrix-core tiles
OUT=gpurun_out/${1:-forms}; mkdir -p $OUT
VARS=${VARS:-"1024 256"}
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest_gpu.log
fi
for v in $VARS; do
echo "== variant $v"
QPX_VARIANT=$v timeout 300 python scripts/prof_phases.py 512 100 100 0 > $OUT/phases_c2_f64_v$v.log 2>&1; tail -9 $OUT/phases_c2_f64_v$v.log
QPX_VARIANT=$v timeout 300 python scripts/prof_phases.py 4096 64 64 0 > $OUT/phases_c5_v$v.log 2>&1; tail -9 $OUT/phases_c5_v$v.log
QPX_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_v$v.json 2> $OUT/bench_v$v.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_v$v.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['ipm_iterations_mean'])"
QPX_VARIANT=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --batch 8192 --nz 64 --nineq 64 > $OUT/bench_c5_v$v.json 2> $OUT/bench_c5_v$v.err; python -c "
import json; d=json.load(open('$OUT/bench_c5_v$v.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['ipm_iterations_mean'])"
done
