#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, -m gpu tests, bench, rocprofv3 kernel stats + HBM PMC.
# Everything is bounded by `timeout`; outputs land in gpurun_out/$TAG.  The *.txt summaries under
# gpurun_out/$TAG/profiles are what gets copied into profiles/ (tracked).
TAG=${1:-r01}
OUT=gpurun_out/$TAG
PROF=$OUT/profiles
mkdir -p $OUT $PROF
export TMPDIR=/tmp
REPO=$(pwd)
echo "== env" | tee $OUT/summary.txt
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)") >> $OUT/summary.txt 2>&1
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log >> $OUT/summary.txt
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest_gpu.log >> $OUT/summary.txt
fi
echo "== bench" | tee -a $OUT/summary.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -5 $OUT/bench.err >> $OUT/summary.txt
timeout 300 python bench.py --dtype f32 --no-cpu-baseline > $OUT/bench_f32.json 2>> $OUT/bench.err
cat $OUT/bench_f32.json >> $OUT/summary.txt
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats -- $CMD > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $?" | tee -a $OUT/summary.txt
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline"; echo "# bench line of that run:"; grep '^{' $OUT/prof_stats.log | sed 's/^/# /';
  find /tmp/prof_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_kernel_stats.txt 2>&1
cat $PROF/${TAG}_kernel_stats.txt >> $OUT/summary.txt
echo "== rocprofv3 pmc" | tee -a $OUT/summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $?" | tee -a $OUT/summary.txt
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/${TAG}_pmc_$C.txt 2>&1
  cat $PROF/${TAG}_pmc_$C.txt >> $OUT/summary.txt
done
python scripts/make_traffic_json.py $PROF/${TAG}_pmc_FETCH_SIZE.txt $PROF/${TAG}_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json 2>> $OUT/summary.txt
cat $PROF/ipm_traffic.json >> $OUT/summary.txt
du -sh $OUT | tee -a $OUT/summary.txt
# same-box A/B against reference builds, when they travel with the snapshot (qpth_amd/libqpx_hip_<tag>.so)
if ls qpth_amd/libqpx_hip_r02q.so > /dev/null 2>&1; then
  echo "== A/B on this box (r02q = the previous measured build, p4 = four-column panels)" | tee -a $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py $(ls qpth_amd/libqpx_hip_p4.so qpth_amd/libqpx_hip_r02q.so 2>/dev/null) qpth_amd/libqpx_hip.so 2>&1 | grep -v amdgpu.ids | tee $PROF/${TAG}_ab_c2.txt >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02q.so qpth_amd/libqpx_hip.so 8192 64 64 0 2>&1 | grep -v amdgpu.ids | tee $PROF/${TAG}_ab_c5shape.txt >> $OUT/summary.txt
  timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r02q.so qpth_amd/libqpx_hip.so 512 100 50 10 2>&1 | grep -v amdgpu.ids | tee $PROF/${TAG}_ab_c3.txt >> $OUT/summary.txt
fi
