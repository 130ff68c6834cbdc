#!/bin/bash
# GPU visit: large-QP family A/B of knob values at C4 (bench lines + per-dispatch timeline of the default)
TAG=${1:-r02h}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -s -k "c4 or large_qp" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee $OUT/summary.txt
tail -3 $OUT/pytest_gpu.log >> $OUT/summary.txt
for V in "$@"; do
  echo "== bench c4 QPX_VARIANT=$V" >> $OUT/summary.txt
  QPX_VARIANT=$V timeout 300 python bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step %.3f  QPs/s %.1f  kernel_ms %s  fwd_only_ms %.3f roofline %s' % (d['ms_per_step'], d['value'], json.dumps(d['kernel_ms']), d['fwd_only']['ms'], json.dumps({k: d['roofline'][k] for k in ('achieved','frac','launch_ms')})))
" >> $OUT/summary.txt
done
V=$1
(cd /tmp && QPX_VARIANT=$V timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4_$V -o stats -- python $REPO/bench.py --config c4 --steps 3 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_c4_$V.log 2>&1); echo "rocprof $V exit $?" | tee -a $OUT/summary.txt
find /tmp/prof_c4_$V -name "*.db" | while read f; do python scripts/rocprof_timeline.py "$f" --last 900; python scripts/rocprof_summary.py "$f"; done > $OUT/c4_timeline_$V.txt 2>&1
find /tmp/prof_c4_$V -name "*.db" | while read f; do python scripts/rocprof_timeline.py "$f" --last 450 --list; done > $OUT/c4_timeline_list_$V.txt 2>&1
