#!/bin/bash
# Round 4, visit a: the pipelined GEMM tile kernel of the large-QP family against the round-3 one (knob bit 30) on the
# same box, a per-launch timeline of the C4 forward, and the large-family parity tests.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
: > $OUT/summary.txt
echo "== parity of the large-QP family (pipelined GEMM)" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "c4 or large_qp or refused or every_loop_kernel_form" > $OUT/pytest.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest.log >> $OUT/summary.txt
echo "== A/B v1 (bit 30) vs pipelined" | tee -a $OUT/summary.txt
for dims in "128 500 500 0" "64 300 300 0"; do
  echo "-- B n m q = $dims" >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:1073741824 qpth_amd/libqpx_hip.so:0 $dims 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
echo "($(el))" | tee -a $OUT/summary.txt
echo "== C4 bench line" | tee -a $OUT/summary.txt
timeout 300 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench.err
cut -c1-1800 $OUT/bench_c4.json >> $OUT/summary.txt
echo "== timeline of the C4 forward" | tee -a $OUT/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python $REPO/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_tl.log 2>&1); echo "rocprof exit $? ($(el))" | tee -a $OUT/summary.txt
find /tmp/prof_tl -name "*.db" | while read f; do python scripts/rocprof_timeline.py "$f" --last 800 > $OUT/timeline.txt 2>&1; python scripts/rocprof_timeline.py "$f" --last 420 --list > $OUT/timeline_list.txt 2>&1; done
head -50 $OUT/timeline.txt >> $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
