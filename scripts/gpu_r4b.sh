#!/bin/bash
# Round 4, visit b: the large-QP family with equality constraints, float32 tensors in float64 arithmetic, the
# chain-wave diagonal blocks, sixteen-wave substitutions and the mat-vec beside the factorisation -- parity, then
# same-box A/B of every new piece against its round-3 form (knob bits 25, 26, 27, 30), then the dispatch question
# (workgroup kernels, knob 1, against the large-QP family, knob 3, at sizes both serve).
TAG=${1:-r04b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
: > $OUT/summary.txt
echo "== parity" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -s -k "c4 or large_qp or refused or every_loop_kernel_form or float32 or golden_batches" > $OUT/pytest.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
grep -a "rel err" $OUT/pytest.log | sort -u >> $OUT/summary.txt
tail -6 $OUT/pytest.log >> $OUT/summary.txt
echo "== A/B at C4: all new | 4-wave substitutions | mat-vec in front | one-wave diagonal blocks | round-3 GEMM + diag" | tee -a $OUT/summary.txt
timeout 400 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:0 qpth_amd/libqpx_hip.so:33554432 qpth_amd/libqpx_hip.so:67108864 qpth_amd/libqpx_hip.so:134217728 qpth_amd/libqpx_hip.so:1073741824 qpth_amd/libqpx_hip.so:1174405120 128 500 500 0 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
echo "($(el))" | tee -a $OUT/summary.txt
echo "== C4 with equality constraints (128 500 400 100), C4 shape at 64 QPs" | tee -a $OUT/summary.txt
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:0 128 500 400 100 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT/summary.txt
echo "== dispatch: workgroup kernels (1) vs large-QP family (3)" | tee -a $OUT/summary.txt
for dims in "512 150 150 0" "512 120 120 30" "128 190 190 0" "2048 150 150 0"; do
  echo "-- B n m q = $dims" >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:1 qpth_amd/libqpx_hip.so:3 $dims 2>&1 | grep -v amdgpu.ids | tail -2 >> $OUT/summary.txt
done
echo "($(el))" | tee -a $OUT/summary.txt
echo "== C4 bench lines (f64, f32 tensors)" | tee -a $OUT/summary.txt
timeout 300 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench.err
cut -c1-1900 $OUT/bench_c4.json >> $OUT/summary.txt
timeout 300 python bench.py --config c4 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c4_f32.json 2>> $OUT/bench.err
cut -c1-900 $OUT/bench_c4_f32.json >> $OUT/summary.txt
tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== timeline of the C4 forward" | tee -a $OUT/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python $REPO/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_tl.log 2>&1); echo "rocprof exit $? ($(el))" | tee -a $OUT/summary.txt
find /tmp/prof_tl -name "*.db" | while read f; do python scripts/rocprof_timeline.py "$f" --last 800 > $OUT/timeline.txt 2>&1; python scripts/rocprof_timeline.py "$f" --last 420 --list > $OUT/timeline_list.txt 2>&1; done
head -45 $OUT/timeline.txt >> $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
