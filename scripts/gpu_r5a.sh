#!/bin/bash
# Round 5, visit a: (1) where the two workgroups of a CU land and what HW_ID.tg_id says; (2) the forward as one launch
# against two, each with the CU's second workgroup started late (QPX_TUNE_DEPHASE), C2 and C5's shape; (3) large-QP family:
# substitutions on the lower triangle + prefetch, no mirrored panels -- same-box A/B against the round-4 library, parity
# tests of that family, kernel stats; (4) bench lines C2 / C4.
TAG=${1:-r05a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
S=$OUT/summary.txt
echo "== env" > $S
(rocm-smi --showproductname 2>/dev/null | head -6; nproc; python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))") >> $S 2>&1
echo "== placement probe" >> $S
timeout 200 python scripts/probe_simd.py --placement-only 2>&1 | grep -v amdgpu.ids > $OUT/probe.txt; head -30 $OUT/probe.txt >> $S
echo "== forward: one launch vs two, dephase (C2)" >> $S
timeout 300 python scripts/ab_fwd.py 512 100 100 0 0 1 2 3 4 6 8 2>&1 | grep -v amdgpu.ids > $OUT/ab_fwd_c2.txt; cat $OUT/ab_fwd_c2.txt >> $S
echo "== forward: one launch vs two, dephase (B=8192 n=m=64)" >> $S
timeout 300 python scripts/ab_fwd.py 8192 64 64 0 0 2 4 2>&1 | grep -v amdgpu.ids > $OUT/ab_fwd_c5shape.txt; cat $OUT/ab_fwd_c5shape.txt >> $S
echo "== forward: one launch vs two, dephase (B=2048 n=m=100)" >> $S
timeout 300 python scripts/ab_fwd.py 2048 100 100 0 0 2 4 2>&1 | grep -v amdgpu.ids > $OUT/ab_fwd_b2048.txt; cat $OUT/ab_fwd_b2048.txt >> $S
echo "== C4 A/B against the round-4 library" >> $S
timeout 400 python scripts/ab_bench.py qpth_amd/libqpx_hip_r04.so qpth_amd/libqpx_hip.so 128 500 500 0 2>&1 | grep -v amdgpu.ids > $OUT/ab_c4.txt; cat $OUT/ab_c4.txt >> $S
timeout 300 python scripts/ab_bench.py qpth_amd/libqpx_hip_r04.so qpth_amd/libqpx_hip.so 512 150 150 0 2>&1 | grep -v amdgpu.ids > $OUT/ab_n150.txt; cat $OUT/ab_n150.txt >> $S
echo "== pytest (large-QP family, one-launch forward, accuracy options)" >> $S
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -s -k "full_size_matches_oracle_c4 or large_qps_with_equality or accuracy_options or one_launch or c4_float32 or refinement_is_refused or solver_entry_points" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $S
grep -v amdgpu.ids $OUT/pytest.log | tail -25 >> $S
echo "== bench" >> $S
timeout 600 python bench.py > $OUT/bench_c2.json 2> $OUT/bench.err; echo "bench c2 exit $?" >> $S; cat $OUT/bench_c2.json >> $S
timeout 600 python bench.py --config c4 --steps 20 --warmup 3 > $OUT/bench_c4.json 2>> $OUT/bench.err; echo "bench c4 exit $?" >> $S; cat $OUT/bench_c4.json >> $S
echo "== rocprofv3 kernel stats C4" >> $S
CMD="python $REPO/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o stats -- $CMD > $REPO/$OUT/prof_c4.log 2>&1); echo "rocprof exit $?" >> $S
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline"; grep '^{' $OUT/prof_c4.log | sed 's/^/# /' | cut -c1-600;
  find /tmp/prof_c4 -name "*.db" | while read f; do python scripts/rocprof_summary.py "$f"; done; } > $OUT/${TAG}_c4_kernel_stats.txt 2>&1
cat $OUT/${TAG}_c4_kernel_stats.txt >> $S
tail -5 $OUT/bench.err >> $S
du -sh $OUT >> $S
