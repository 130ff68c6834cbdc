#!/bin/bash
# Round 4, visit d: the panels finished inside the update launches (flag hand-over between workgroups of one launch):
# large-family parity on the hardware, then same-box A/B against the round's previous order (knob bit 24).
TAG=${1:-r04d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
: > $OUT/summary.txt
echo "== parity (large-QP family)" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "c4 or large_qp or refused or every_loop_kernel_form or equality" > $OUT/pytest.log 2>&1; echo "pytest exit $? ($(el))" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest.log >> $OUT/summary.txt
echo "== A/B: panels inside the update launches (0) | in launches of their own (bit 24)" | tee -a $OUT/summary.txt
for dims in "128 500 500 0" "512 150 150 0" "128 500 400 100"; do
  echo "-- B n m q = $dims" >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:0 qpth_amd/libqpx_hip.so:16777216 $dims 2>&1 | grep -v amdgpu.ids >> $OUT/summary.txt
done
echo "($(el))" | tee -a $OUT/summary.txt
python - <<'PY' >> $OUT/summary.txt 2>&1
# status words after a C4 forward: no QP may carry a failure bit (a timed-out flag wait sets KKT_BREAKDOWN)
import sys, numpy as np, torch
sys.path.insert(0, "tests")
import problems
from qpth_amd.kkt import KKTFactors
dev = torch.device("cuda:0")
Q, p, G, h, A, b = [torch.tensor(x, device=dev) for x in problems.prof_qp(128, 500, 500, 0, 0)]
fac = KKTFactors.build(Q, G, A)
for rep in range(20):
    res = fac.ipm(p, h, b)
torch.cuda.synchronize()
print("status bits after 20 C4 forwards:", int(res.status.max().item()), "iterations mean %.2f" % res.iters.float().mean().item())
PY
du -sh $OUT | tee -a $OUT/summary.txt
