#!/bin/bash
# Round 6, first visit: (1) the RCCL code path with ONE rank on the one GPU of the box (tests + bench.py's distributed branch),
# (2) the per-phase tick ledger of the C2 loop kernel refreshed on the round-5 kernels (profiling builds of the same
# sources: libqpx_hip_prof.so / libqpx_hip_pprof.so), (3) the default bench line with its new fields.
TAG=${1:-r06a}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
S=$OUT/summary.txt
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== smoke" > $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $? ($(el))" >> $S
tail -2 $OUT/smoke.log >> $S
echo "== RCCL, world size 1" >> $S
timeout 900 python -m pytest tests/test_gpu_dist.py -q --timeout 800 -s -k "world_size_one" > $OUT/pytest_dist.log 2>&1; echo "pytest exit $? ($(el))" >> $S
grep -v amdgpu.ids $OUT/pytest_dist.log | tail -25 >> $S
cp gpurun_out/nccl_world1_bench.json $PROF/${TAG}_nccl_world1_bench.json 2>/dev/null
echo "== phase ledger (profiling build)" >> $S
for dims in "512 100 100 0" "256 100 100 0"; do
  timeout 300 python scripts/prof_phases.py $dims 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_phases_c2.txt
  timeout 300 python scripts/prof_panel.py $dims 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_phases_c2.txt
done
timeout 300 python scripts/prof_phases.py 8192 64 64 0 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_phases_c5shape.txt
cat $PROF/${TAG}_phases_c2.txt $PROF/${TAG}_phases_c5shape.txt >> $S
echo "($(el))" >> $S
echo "== bench (default)" >> $S
timeout 600 python bench.py > $PROF/${TAG}_bench_f64.json 2> $OUT/bench.err; echo "bench exit $? ($(el))" >> $S
cat $PROF/${TAG}_bench_f64.json >> $S
tail -5 $OUT/bench.err >> $S
echo "== shared-parameter gradients (ABI v8 batch mean) + a few parity tests" >> $S
timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 800 -x -k "shared or golden or smoke or c1" > $OUT/pytest_some.log 2>&1; echo "pytest exit $? ($(el))" >> $S
grep -v amdgpu.ids $OUT/pytest_some.log | tail -8 >> $S
du -sh $OUT >> $S
