#!/bin/bash
# Round 4, visit o (short): SQ counters of the matrix-core pre-factorisation (where do its wave cycles go?)
TAG=${1:-r04o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_WAIT|SQ_ACTIVE_INST_ANY|SQ_INST_CYCLES|SQ_WAVE_CYCLES" | cut -c1-160 | sort -u | head -60 > $R/$OUT/avail.txt
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d $R/$OUT/pmc_$tag -o pmc -- python $R/scripts/run_prefac.py 512 100 100 6 > $R/$OUT/pmc_$tag.log 2>&1
  echo "pass $tag exit $?" >> $R/$OUT/summary.txt
done
cd $R
OUTDIR=$OUT python - <<'PY' >> $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04o"
import os
out = os.environ.get("OUTDIR", out)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pmc_*/**/*counter_collection*.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")[:40]
        a = acc[k][row.get("Counter_Name")]
        a[0] += float(row.get("Counter_Value", 0)); a[1] += 1
for k, d in acc.items():
    if "prefac" in k or "sweep" in k:
        print(k)
        for c, (v, n) in sorted(d.items()):
            print("   %-28s %16.0f per dispatch (%d dispatches)" % (c, v / max(n, 1), n))
PY
cat $OUT/summary.txt
