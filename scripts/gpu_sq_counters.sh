#!/bin/bash
# SQ counters of the loop kernel (rocprofv3 --pmc passes + --kernel-trace only): gpu_sq_counters.sh TAG "B n m q" [QPX_VARIANT]
TAG=$1; DIMS=$2; export QPX_VARIANT=${3:-0}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
F=$OUT/${TAG}_sq_counters.txt
echo "# rocprofv3 --kernel-trace --pmc <set> -- python scripts/run_ipm.py $DIMS 3   (QPX_VARIANT=$QPX_VARIANT)" > $F
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/sq_$TAG_$i -o pmc -- python $REPO/scripts/run_ipm.py $DIMS 3 > $REPO/$OUT/sq_$i.log 2>&1)
  find /tmp/sq_$TAG_$i -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f" | grep "k_ipm" | grep "SQ_" >> $F; python scripts/rocprof_summary.py "$f" | grep "k_ipm" | head -1 >> $OUT/kernels.txt; done
done
cat $F
