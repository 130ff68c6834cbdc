#!/bin/bash
# Round 6, closing visit on the final sources (after gpu_r6_final.sh's full visit r06z, one change: the large-QP family's
# backward runs one part): smoke, the default bench line, rocprofv3 kernel stats + the two HBM PMC passes of the SAME digest
# (-> ipm_traffic.json), the default line once more (now with `traffic`), the large-QP tests and C4 lines.
TAG=${1:-r06zz}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
REPO=$(pwd)
S=$OUT/summary.txt
date +%s > $OUT/t0
el() { echo "$(( $(date +%s) - $(cat $OUT/t0) )) s"; }
echo "== smoke" > $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $? ($(el))" >> $S
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --step-kernels-only"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- $CMD > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $? ($(el))" >> $S
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --step-kernels-only"; echo "# bench line of that run:"; grep '^{' $OUT/prof_stats.log | sed 's/^/# /';
  find /tmp/prof_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs --step-kernels-only > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $? ($(el))" >> $S
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs --step-kernels-only"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_pmc_$C.txt 2>&1
done
python scripts/make_traffic_json.py $PROF/${TAG}_pmc_FETCH_SIZE.txt $PROF/${TAG}_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json 2>> $S
cp $PROF/ipm_traffic.json profiles/ipm_traffic.json
echo "== bench (default; the PMC record of this digest in place)" >> $S
timeout 400 python bench.py > $PROF/${TAG}_bench_f64.json 2> $OUT/bench.err; echo "bench exit $? ($(el))" >> $S
cat $PROF/${TAG}_bench_f64.json >> $S
timeout 200 python bench.py --shared --no-cpu-baseline > $PROF/${TAG}_bench_shared.json 2>> $OUT/bench.err
timeout 300 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_c4.json 2>> $OUT/bench.err
timeout 300 python bench.py --config custom --batch 128 --nz 500 --nineq 400 --neq 100 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_c4_neq100.json 2>> $OUT/bench.err
timeout 300 python bench.py --config custom --batch 512 --nz 150 --nineq 150 --neq 0 --steps 20 --warmup 3 --no-cpu-baseline > $PROF/${TAG}_bench_b512_n150_m150.json 2>> $OUT/bench.err
for f in shared c4 c4_neq100 b512_n150_m150; do echo "-- $f" >> $S; cut -c1-330 $PROF/${TAG}_bench_$f.json >> $S; grep -o '"kernel_ms": {[^}]*}' $PROF/${TAG}_bench_$f.json >> $S; done
echo "== the large-QP family's tests + graph capture + RCCL world size 1" >> $S
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "large or c4 or captured_graph or caller_stream or world_size_one or external or kkt" > $OUT/pytest.log 2>&1; echo "pytest exit $? ($(el))" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -4 >> $S
cp gpurun_out/nccl_world1_bench.json $PROF/${TAG}_nccl_world1_bench.json 2>/dev/null
for dims in "128 500 500 0" "512 150 150 0"; do
  echo "-- B n m q = $dims" >> $PROF/${TAG}_ab_r05.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r05.so qpth_amd/libqpx_hip.so $dims 2>&1 | grep -v amdgpu.ids >> $PROF/${TAG}_ab_r05.txt
done
cat $PROF/${TAG}_ab_r05.txt >> $S
