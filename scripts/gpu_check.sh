#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, -m gpu tests, bench, rocprofv3 kernel stats + HBM PMC.
# Everything is bounded by `timeout`; outputs land in gpurun_out/$TAG.
TAG=${1:-r1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== env" | tee $OUT/summary.txt
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)") >> $OUT/summary.txt 2>&1
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log >> $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -15 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt; tail -5 $OUT/bench.err >> $OUT/summary.txt
timeout 300 python bench.py --steps 20 --warmup 3 --dtype f32 --no-cpu-baseline > $OUT/bench_f32.json 2>> $OUT/bench.err
cat $OUT/bench_f32.json >> $OUT/summary.txt
echo "== rocprofv3 kernel stats" | tee -a $OUT/summary.txt
REPO=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof_stats -o stats -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $REPO/$OUT/prof_stats.log 2>&1); echo "rocprof stats exit $?" | tee -a $OUT/summary.txt
find $OUT/prof_stats -name "*kernel_stats*.csv" | head -2 | while read f; do head -12 "$f" >> $OUT/summary.txt; done
echo "== rocprofv3 pmc" | tee -a $OUT/summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -d $REPO/$OUT/prof_pmc_$C -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/$OUT/prof_pmc_$C.log 2>&1); echo "pmc $C exit $?" | tee -a $OUT/summary.txt
done
python scripts/summarize_pmc.py $OUT >> $OUT/summary.txt 2>&1
# keep the merged output small
find $OUT -name "*.db" -size +20M -delete
du -sh $OUT | tee -a $OUT/summary.txt
