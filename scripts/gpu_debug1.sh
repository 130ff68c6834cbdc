#!/bin/bash
OUT=gpurun_out/${1:-dbg1}; mkdir -p $OUT
timeout 600 python scripts/stress_spd.py > $OUT/stress.log 2>&1; echo "stress exit $?"; cat $OUT/stress.log | tail -30
timeout 300 python scripts/prof_phases.py 512 100 100 0 > $OUT/phases_c2_f64.log 2>&1; cat $OUT/phases_c2_f64.log | tail -12
timeout 300 python scripts/prof_phases.py 512 100 100 0 f32 > $OUT/phases_c2_f32.log 2>&1; cat $OUT/phases_c2_f32.log | tail -12
timeout 300 python scripts/prof_phases.py 4096 64 64 0 > $OUT/phases_c5.log 2>&1; cat $OUT/phases_c5.log | tail -12
timeout 300 python scripts/prof_phases.py 256 100 100 0 > $OUT/phases_c2_b256.log 2>&1; cat $OUT/phases_c2_b256.log | tail -12
