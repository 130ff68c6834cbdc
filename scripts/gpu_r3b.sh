#!/bin/bash
# Round 3, visit b: the chain-wave form of the four-wave tile kernels.  Parity subset first, then same-box A/B against
# the round-2 library (libqpx_hip_r02.so) and against this build without the chain wave (variant + 16384).
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== pytest -m gpu, subset ($(el))" | tee $OUT/summary.txt
timeout 400 python -m pytest tests -m gpu -q -x --timeout 200 -k "golden_batches or against_oracle or full_size_matches_oracle_c2 or every_loop_kernel_form or solver_entry or hard_problems or edge" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== A/B C2 (B=512): round-2 library | this build, chain wave | this build, no chain wave ($(el))" | tee -a $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so qpth_amd/libqpx_hip.so:16384 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c2.txt >> $OUT/summary.txt
echo "== A/B one QP per CU (B=256) ($(el))" | tee -a $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so qpth_amd/libqpx_hip.so:16384 256 100 100 0 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c2_b256.txt >> $OUT/summary.txt
echo "== A/B full chip (B=2048, nz=nineq=100) ($(el))" | tee -a $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so qpth_amd/libqpx_hip.so:16384 2048 100 100 0 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_b2048.txt >> $OUT/summary.txt
echo "== A/B C3 / C5 shapes (no chain form there; the laundered addresses) ($(el))" | tee -a $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so 512 100 50 10 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c3.txt >> $OUT/summary.txt
timeout 120 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02.so qpth_amd/libqpx_hip.so 8192 64 64 0 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c5shape.txt >> $OUT/summary.txt
echo "== bench ($(el))" | tee -a $OUT/summary.txt
timeout 200 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== done ($(el))" | tee -a $OUT/summary.txt
