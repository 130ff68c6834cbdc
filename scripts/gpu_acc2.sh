#!/bin/bash
mkdir -p gpurun_out/acc2
python scripts/rcp_check.py 2>&1 | tail -2
echo "--- default build"; python scripts/gpu_accuracy_probe.py 2>&1 | grep -E "variant 0|factor_solve" | tail -3
echo "--- exact reciprocal build"; cp qpth_amd/libqpx_hip.so /tmp/keep.so; cp qpth_amd/libqpx_hip_exact.so qpth_amd/libqpx_hip.so; python scripts/gpu_accuracy_probe.py 2>&1 | grep -E "variant 0|factor_solve" | tail -3; cp /tmp/keep.so qpth_amd/libqpx_hip.so
