#!/usr/bin/env python3
"""The 16 x 16 pivot block of the tile kernels' sixteen-column panels, timed alone and next to a second wave on the
same SIMD that streams f64 MFMAs / f64 FMAs / f32 FMAs (libqpx_bench.so, `make -C qpth_amd/csrc bench`)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qpth_amd", sys.argv[1] if len(sys.argv) > 1 else "libqpx_bench.so"))
lib.qpx_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
out = torch.zeros(20480 + 4096, dtype=torch.float64, device=dev)
inp = torch.rand(4096, dtype=torch.float64, device=dev) * 0.5 + 0.75
names = {30: "alone (one wave per workgroup)", 31: "partner wave on the same SIMD streams f64 MFMAs",
         32: "partner streams f64 vector FMAs", 33: "partner streams f32 vector FMAs"}
for blocks in (1, 256, 1024):
    print("== %d workgroups" % blocks)
    for which in (30, 31, 32, 33):
        out.zero_()
        rc = lib.qpx_bench(which, blocks, 100, 0, out.data_ptr(), inp.data_ptr(), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        o = out.cpu().numpy()[4096:4096 + blocks]
        print("  %-52s ticks per pivot block: mean %8.0f  min %8.0f  max %8.0f   (%.0f per pivot)" % (names[which], o.mean(), o.min(), o.max(), o.mean() / 16))
