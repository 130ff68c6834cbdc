#!/bin/bash
# Round 6: same-box A/B of the working tree's library against the archived round-5 build (qpth_amd/libqpx_hip_r05.so) at the
# shapes the tile kernels serve, + the GPU parity tests of the loop kernel.  TAG = record name.
TAG=${1:-r06b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
S=$OUT/summary.txt
: > $S
for dims in "512 100 100 0" "512 100 50 10" "8192 64 64 0" "256 100 100 0"; do
  echo "-- B n m q = $dims" >> $OUT/${TAG}_ab_r05.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r05.so qpth_amd/libqpx_hip.so $dims 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_ab_r05.txt
done
cat $OUT/${TAG}_ab_r05.txt >> $S
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --timeout 900 -x -k "${TESTS:-not large and not c4 and not big}" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -12 >> $S
fi
