#!/bin/bash
# f64 matrix-instruction micro-benchmark incl. the 4x4x4 (four blocks) form; then the two PMC passes -> ipm_traffic.json
OUT=gpurun_out/r03y
PROF=$OUT/profiles
mkdir -p $PROF
export TMPDIR=/tmp
REPO=$(pwd)
timeout 120 python scripts/ubench_mfma.py 2>&1 | grep -v amdgpu.ids > $PROF/r03y_mfma_ubench.txt
cat $PROF/r03y_mfma_ubench.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $?"
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    find /tmp/prof_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $PROF/r03y_pmc_$C.txt 2>&1
done
python scripts/make_traffic_json.py $PROF/r03y_pmc_FETCH_SIZE.txt $PROF/r03y_pmc_WRITE_SIZE.txt > $PROF/ipm_traffic.json
cat $PROF/ipm_traffic.json
timeout 200 python bench.py --no-cpu-baseline | cut -c1-1500
