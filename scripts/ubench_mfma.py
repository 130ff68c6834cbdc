#!/usr/bin/env python3
"""f64 MFMA micro-benchmark (libqpx_bench.so): clock64 ticks per v_mfma_f64_16x16x4_f64 with 8, 2 and 1
accumulators in flight, against the same rank-4 tile update with 16 vector FMAs per lane."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qpth_amd", "libqpx_bench.so"))
lib.qpx_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
out = torch.zeros(20480 + 4096, dtype=torch.float64, device=dev)
inp = torch.rand(4096, dtype=torch.float64, device=dev) * 0.5 + 0.75
names = {20: "mfma f64 16x16x4, 8 accumulators", 22: "mfma f64 16x16x4, 2 accumulators", 21: "mfma f64 16x16x4, dependent chain",
         23: "rank-4 tile update by 16 vector FMAs", 24: "mfma f64 4x4x4 (4 blocks), 8 accumulators",
         25: "mfma f64 4x4x4 (4 blocks), dependent chain"}
for blocks in (1, 512, 1024, 2048):
    for which in (20, 22, 21, 23, 24, 25):
        out.zero_()
        assert lib.qpx_bench(which, blocks, 500, 0, out.data_ptr(), inp.data_ptr(), None) == 0
        torch.cuda.synchronize()
        t = out[4096:4096 + blocks].cpu().numpy()
        print("%5d waves  %-42s ticks per instruction (23: per tile update): mean %7.1f min %7.1f max %7.1f" % (blocks, names[which], t.mean(), t.min(), t.max()))
