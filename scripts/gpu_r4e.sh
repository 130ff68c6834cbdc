#!/bin/bash
# Round 4, visit e: block-major storage of the large-QP family's matrices and the panel hand-over with relaxed polling,
# against the row-major build of the previous visit (qpth_amd/libqpx_hip_r04d.so, knob bit 24 = panels in launches of
# their own = the r04c order) on the same box.
TAG=${1:-r04e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
: > $OUT/summary.txt
echo "== parity (large-QP family)" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "full_size_matches_oracle_c4 or equality or large_qp_hbm or (every_loop_kernel_form and 3-)" > $OUT/pytest.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest.log >> $OUT/summary.txt
echo "== A/B: row-major, panels in own launches (r04c order) | block-major + panels inside the update launches | block-major, panels in own launches" | tee -a $OUT/summary.txt
for dims in "128 500 500 0" "512 150 150 0" "128 500 400 100"; do
  echo "-- B n m q = $dims" >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r04d.so:16777216 qpth_amd/libqpx_hip.so:0 qpth_amd/libqpx_hip.so:16777216 $dims 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
echo "($(el))" | tee -a $OUT/summary.txt
echo "== timeline of the C4 forward (default build)" | tee -a $OUT/summary.txt
REPO=$(pwd)
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python $REPO/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_tl.log 2>&1); echo "rocprof exit $? ($(el))" | tee -a $OUT/summary.txt
find /tmp/prof_tl -name "*.db" | while read f; do python scripts/rocprof_timeline.py "$f" --last 800 > $OUT/timeline.txt 2>&1; done
head -40 $OUT/timeline.txt >> $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
