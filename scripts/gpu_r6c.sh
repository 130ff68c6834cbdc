#!/bin/bash
TAG=${1:-r06c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
S=$OUT/summary.txt
: > $S
echo "== phase ledger of the ahead loop (profiling build)" >> $S
for dims in "512 100 100 0" "256 100 100 0"; do
  timeout 300 python scripts/prof_phases.py $dims 2>&1 | grep -v amdgpu.ids | grep -v "k_sweep\|load Q\|G^T 1\|sweep n\|scatter" >> $OUT/${TAG}_phases.txt
done
cat $OUT/${TAG}_phases.txt >> $S
echo "== occupancy probe, C5 shape" >> $S
timeout 600 python scripts/occupancy_probe.py 65536 64 64 0 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_occupancy_c5.txt
QPX_VARIANT=8192 PADS="0 12000 40000" timeout 600 python scripts/occupancy_probe.py 65536 64 64 0 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_occupancy_c5.txt
PADS="0 10000 90000" timeout 600 python scripts/occupancy_probe.py 2048 100 100 0 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_occupancy_c5.txt
cat $OUT/${TAG}_occupancy_c5.txt >> $S
timeout 600 python -m pytest tests/test_gpu_parity.py -q --timeout 500 -x -k "captured_graph" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -5 >> $S
