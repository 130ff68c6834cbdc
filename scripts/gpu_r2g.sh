#!/bin/bash
# GPU visit: large-QP family with the batch split over side streams (knob bits 16..27) -- A/B at C4, parity, timeline
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -s -k "c4 or large_qp" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log >> $OUT/summary.txt
for V in 65536 262144 131072 196608 $((262144 + (6<<20))) $((262144 + (20<<20))) $((131072 + (12<<20))); do
  echo "== bench c4 QPX_VARIANT=$V (parts $((V>>16 & 15)), stagger $(( (V>>20) * 16 )) us)" >> $OUT/summary.txt
  QPX_VARIANT=$V timeout 300 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step %.3f  QPs/s %.1f  kernel_ms %s  fwd_only_ms %.3f' % (d['ms_per_step'], d['value'], json.dumps(d['kernel_ms']), d['fwd_only']['ms']))
" >> $OUT/summary.txt
done
for V in 65536 262144; do
  (cd /tmp && QPX_VARIANT=$V timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4_$V -o stats -- python $REPO/bench.py --config c4 --steps 3 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_c4_$V.log 2>&1); echo "rocprof $V exit $?" | tee -a $OUT/summary.txt
  find /tmp/prof_c4_$V -name "*.db" | while read f; do python scripts/rocprof_timeline.py "$f" --last 900; done > $OUT/c4_timeline_$V.txt 2>&1
  find /tmp/prof_c4_$V -name "*.db" | while read f; do python scripts/rocprof_timeline.py "$f" --last 450 --list; done > $OUT/c4_timeline_list_$V.txt 2>&1
done
