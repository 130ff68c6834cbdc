#!/bin/bash
# after the chain-wave form became the default at four tile rows: bench lines of the configurations it serves (C3, C5 on one GPU)
# + the default line (box check: k_sweep ran 0.247 ms instead of 0.157 on the box of the visit before)
OUT=gpurun_out/r03zc
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --no-cpu-baseline > $OUT/r03z_bench_f64_boxcheck.json 2> $OUT/err.txt
timeout 200 python bench.py --config c3 --no-cpu-baseline > $OUT/r03z_bench_c3.json 2>> $OUT/err.txt
timeout 300 python bench.py --config c5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r03z_bench_c5_one_gpu.json 2>> $OUT/err.txt
for f in f64_boxcheck c3 c5_one_gpu; do python - <<PY
import json
d=json.loads(open("$OUT/r03z_bench_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"]), round(d["ms_per_step"],4), {a:round(b,4) for a,b in d["kernel_ms"].items() if isinstance(b,(int,float))})
PY
done
