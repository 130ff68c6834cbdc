#!/usr/bin/env python3
"""FETCH_SIZE calibration (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`): streams a known byte count once
through 8-B-per-lane buffer loads (the tile loads' pattern) and through 16-B-per-lane global loads."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qpth_amd", "libqpx_bench.so"))
lib.qpx_bench_ptr.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
blocks, m = 2048, 131072                      # 2048 x 1 MiB = 2 GiB, read once (past the 256 MiB Infinity Cache)
src = torch.rand(blocks * m, dtype=torch.float64, device=dev)
out = torch.zeros(4096, dtype=torch.float64, device=dev)
for which in (50, 51, 50, 51):
    assert lib.qpx_bench_ptr(which, blocks, 0, m, out.data_ptr(), src.data_ptr(), None) == 0
    torch.cuda.synchronize()
print("bytes per launch: %d" % (blocks * m * 8))
