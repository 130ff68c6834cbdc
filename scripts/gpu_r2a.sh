#!/bin/bash
# Round-2 first GPU visit: smoke, all -m gpu tests, bench, in-kernel phase/panel timers, SQ instruction-mix counters.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; python -c "import torch;print(torch.__version__, torch.cuda.device_count())") > $OUT/summary.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt
tail -2 $OUT/smoke.log >> $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
grep -E "f32 rel err|passed|failed|FAILED|Error|error" $OUT/pytest_gpu.log | tail -30 >> $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -3 $OUT/bench.err >> $OUT/summary.txt
echo "== phases" >> $OUT/summary.txt
timeout 300 python scripts/prof_phases.py >> $OUT/summary.txt 2>&1
timeout 300 python scripts/prof_panel.py >> $OUT/summary.txt 2>&1
echo "== SQ counters" >> $OUT/summary.txt
(cd /tmp && timeout 120 rocprofv3 -L > $REPO/$OUT/counters_list.txt 2>&1)
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $SET -d /tmp/prof_sq$i -o sq -- $CMD > $REPO/$OUT/prof_sq$i.log 2>&1); echo "sq pass $i exit $?" | tee -a $OUT/summary.txt
  find /tmp/prof_sq$i -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done > $OUT/sq$i.txt 2>&1
  grep -E "k_ipm|k_sweep|k_kkt|counter" $OUT/sq$i.txt >> $OUT/summary.txt
done
du -sh $OUT | tee -a $OUT/summary.txt
