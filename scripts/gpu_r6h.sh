#!/bin/bash
TAG=${1:-r06h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
S=$OUT/summary.txt
: > $S
timeout 300 python scripts/prof_backward.py 2>&1 | grep -v amdgpu.ids >> $S
timeout 300 python scripts/prof_backward.py 512 100 50 10 2>&1 | grep -v amdgpu.ids >> $S
timeout 300 python scripts/prof_prefac.py 2>&1 | grep -v amdgpu.ids >> $S
timeout 300 python scripts/prof_phases.py 512 100 100 0 2>&1 | grep -v amdgpu.ids | grep -v "k_sweep\|load Q\|G^T 1\|sweep n\|scatter" >> $S
bash scripts/gpu_ab_libs.sh $TAG "qpth_amd/libqpx_hip_r05.so qpth_amd/libqpx_hip_v3.so qpth_amd/libqpx_hip.so" "512 100 100 0" "512 100 50 10" >> $S
timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 800 -x -k "golden or c2 or c3 or kkt or backward or external or refine or shared" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -4 >> $S
