#!/bin/bash
TAG=${1:-r06k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
S=$OUT/summary.txt
: > $S
bash scripts/gpu_ab_libs.sh $TAG "qpth_amd/libqpx_hip_v3.so qpth_amd/libqpx_hip.so" "65536 64 64 0" "16384 64 64 0" "65536 32 32 0" "8192 16 16 0" >> $S
PADS="0 6000 14000" timeout 600 python scripts/occupancy_probe.py 65536 64 64 0 2>&1 | grep -v amdgpu.ids >> $S
timeout 300 python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_c5.json 2>> $OUT/bench.err
cut -c1-300 $OUT/${TAG}_bench_c5.json >> $S; grep -o '"kernel_ms": {[^}]*}' $OUT/${TAG}_bench_c5.json >> $S
timeout 1200 python -m pytest tests/test_gpu_parity.py -q --timeout 900 -x -k "c5 or 8192 or one_wave or every_loop_kernel_form or golden or odd or edge or c1" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -4 >> $S
