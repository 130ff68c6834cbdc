#!/bin/bash
OUT=gpurun_out/${1:-dbg4}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest_gpu.log
timeout 300 python scripts/prof_phases.py 512 100 100 0 > $OUT/phases_c2_f64.log 2>&1; cat $OUT/phases_c2_f64.log | tail -10
timeout 300 python scripts/prof_phases.py 512 100 100 0 f32 > $OUT/phases_c2_f32.log 2>&1; cat $OUT/phases_c2_f32.log | tail -10
timeout 300 python scripts/prof_phases.py 4096 64 64 0 > $OUT/phases_c5.log 2>&1; cat $OUT/phases_c5.log | tail -10
timeout 300 python scripts/prof_phases.py 512 100 50 10 > $OUT/phases_c3.log 2>&1; cat $OUT/phases_c3.log | tail -10
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['ipm_iterations_mean'], d['cpu_baseline'])"
