#!/bin/bash
# rocprofv3 kernel stats + the two HBM PMC passes of C5 on one GPU (BASELINE.json configs[4] on one device): gpu_c5_profile.sh TAG
TAG=${1:-r06c5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
CMD="python $REPO/bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --step-kernels-only"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/c5_stats -o stats -- $CMD > $REPO/$OUT/stats.log 2>&1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --step-kernels-only"; grep '^{' $OUT/stats.log | cut -c1-700 | sed 's/^/# /';
  find /tmp/c5_stats -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $OUT/${TAG}_c5_kernel_stats.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/c5_pmc_$C -o pmc -- $CMD > $REPO/$OUT/pmc_$C.log 2>&1)
  { echo "# rocprofv3 --kernel-trace --pmc $C -- python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --step-kernels-only"
    find /tmp/c5_pmc_$C -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $OUT/${TAG}_c5_pmc_$C.txt 2>&1
done
head -12 $OUT/${TAG}_c5_kernel_stats.txt | cut -c1-160; grep "k_ipm\|k_prefac\|k_kkt\|k_sweep" $OUT/${TAG}_c5_pmc_FETCH_SIZE.txt $OUT/${TAG}_c5_pmc_WRITE_SIZE.txt | grep "SIZE " | cut -c1-220
