#!/bin/bash
# Round 5, visit c: the whole -m gpu suite on the sources with the large-QP finishing stage, the dense-solve kernel, the
# two-stage batch contraction and without the round-1 workgroup kernels; large-QP family: parts by default, each with its own
# helper stream (R z' beside the factorisation), against one part on the same box at four shapes; bench lines C2 / C4.
TAG=${1:-r05c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
S=$OUT/summary.txt
echo "== parts (65536 = one part, 0 = automatic, 131072 / 196608 / 262144 = 2 / 3 / 4 parts)" > $S
for dims in "128 500 500 0" "512 150 150 0" "32 500 500 0" "16 300 300 20"; do
  echo "-- B n m q = $dims" >> $S
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip.so:65536 qpth_amd/libqpx_hip.so:0 qpth_amd/libqpx_hip.so:131072 qpth_amd/libqpx_hip.so:196608 qpth_amd/libqpx_hip.so:262144 $dims 2>&1 | grep -v amdgpu.ids | tail -10 >> $S
done
cp $S $OUT/ab_parts.txt
echo "== pytest -m gpu (whole suite)" >> $S
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest_gpu.log | tail -12 >> $S
echo "== bench" >> $S
timeout 600 python bench.py > $OUT/bench_c2.json 2> $OUT/bench.err; echo "bench c2 exit $?" >> $S; cut -c1-900 $OUT/bench_c2.json >> $S
timeout 600 python bench.py --config c4 --steps 20 --warmup 3 > $OUT/bench_c4.json 2>> $OUT/bench.err; echo "bench c4 exit $?" >> $S; cut -c1-400 $OUT/bench_c4.json >> $S
tail -3 $OUT/bench.err >> $S
