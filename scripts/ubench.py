#!/usr/bin/env python3
"""Micro-benchmarks of the gfx950 primitives (libqpx_bench.so): cycles per operation."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "qpth_amd", "libqpx_bench.so"))
lib.qpx_bench.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
out = torch.zeros(20480 + 4096, dtype=torch.float64, device=dev)
inp = torch.rand(4096, dtype=torch.float64, device=dev) * 0.5 + 0.75


def run(which, blocks, reps, m=0, src=None):
    out.zero_()
    rc = lib.qpx_bench(which, blocks, reps, m, out.data_ptr(), (src if src is not None else inp).data_ptr(), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    return o[4096:4096 + blocks], o[8192:8192 + blocks], o[12288:12288 + blocks]


names = {1: "f64 fma independent (cycles/instr)", 2: "f64 fma dependent chain", 3: "ds_read dependent chain",
         4: "LDS publish->consume round trip", 5: "readlane->fma chain (per step)", 6: "rcp_ / full division chain",
         7: "ds_bpermute f64 dependent"}
print("== clock calibration: independent f64 FMAs, long run (reps=20000)")
for blocks in (1, 64, 256, 512, 1024, 2048, 4096):
    a, w, ratio = run(1, blocks, 20000)
    ns_per_instr = w.mean() * 10.0
    print("  %5d waves: clock64 ticks/instr %.2f   wall ns/instr %.3f   clock64 ticks per us %.0f   => if fma issues in 4 shader cycles, shader clock = %.2f GHz" % (
        blocks, a.mean(), ns_per_instr, ratio.mean() * 100, 4.0 / ns_per_instr))
for blocks in (1, 512, 2048):
    print("== %d single-wave workgroups" % blocks)
    for w in (1, 2, 3, 4, 5, 6, 7):
        a, b2, _ = run(w, blocks, 200)
        extra = "   | full division %.1f" % b2.mean() if w == 6 else ""
        print("  %-40s mean %8.1f  min %8.1f max %8.1f%s" % (names[w], a.mean(), a.min(), a.max(), extra))

