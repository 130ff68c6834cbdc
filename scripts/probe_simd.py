#!/usr/bin/env python3
"""Round-3 probes (libqpx_bench.so, `make -C qpth_amd/csrc bench`):
  * where the four waves of 256-thread workgroups land (HW_ID: SIMD / CU / SE, XCC_ID) when two such workgroups share a CU;
  * the 16 x 16 pivot block beside a second pivot chain on the same SIMD, beside an MFMA stream with issue priorities,
    and beside an MFMA stream with a 50 % duty cycle."""
import ctypes
import os
import sys
from collections import Counter, defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qpth_amd", "libqpx_bench.so"))
for f in (lib.qpx_bench, lib.qpx_bench_ptr):
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
out = torch.zeros(65536, dtype=torch.float64, device=dev)
inp = torch.rand(4096, dtype=torch.float64, device=dev) * 0.5 + 0.75

for blocks in (512, 256, 1024):
    out.zero_()
    assert lib.qpx_bench_ptr(40, blocks, 20000, 0, out.data_ptr(), inp.data_ptr(), None) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()[4096:4096 + 16 * blocks].reshape(blocks, 4, 4)
    hw = o[:, :, 0].astype(np.int64)
    xcc = o[:, :, 1].astype(np.int64) & 15
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    t0 = o[:, :, 2]
    distinct = sum(len(set(simd[b])) == 4 for b in range(blocks))
    print("== %d workgroups of 4 waves: %d have their waves on 4 distinct SIMDs" % (blocks, distinct))
    print("   SIMD sequences of waves 0..3 (count):", Counter(tuple(simd[b]) for b in range(blocks)).most_common(8))
    percu = defaultdict(list)
    for b in range(blocks):
        percu[(xcc[b, 0], se[b, 0], sh[b, 0], cu[b, 0])].append(b)
    print("   distinct (xcc, se, sh, cu): %d; workgroups per CU:" % len(percu), Counter(len(v) for v in percu.values()))
    same = tot = 0
    for k, v in percu.items():
        if len(v) == 2:
            tot += 1
            same += int(tuple(simd[v[0]]) == tuple(simd[v[1]]))
    print("   CUs with two workgroups: %d, of which wave w of both on the same SIMD: %d" % (tot, same))
    tg = (hw >> 16) & 15
    print("   HW_ID.tg_id of the workgroups sharing a CU (count):", Counter(tuple(sorted(int(tg[b, 0]) for b in v)) for v in percu.values()).most_common(6))
    print("   workgroup ids sharing a CU (first 6):", [tuple(v) for v in list(percu.values())[:6]])
    print("   first blocks: ", [(b, int(xcc[b, 0]), int(se[b, 0]), int(cu[b, 0]), tuple(int(x) for x in simd[b])) for b in range(12)])
    print("   start spread (100 MHz ticks): %.0f" % (t0.max() - t0.min()))

if '--placement-only' in sys.argv:
    sys.exit(0)
names = {30: "alone", 31: "beside an MFMA stream (no priorities)", 41: "beside a second pivot chain on the same SIMD",
         42: "beside an MFMA stream, pivot at s_setprio 3 / MFMA at 0", 43: "beside a 50 % duty MFMA stream, priorities"}
for blocks in (1, 256):
    print("== pivot block, %d workgroups" % blocks)
    for which in (30, 31, 41, 42, 43):
        out.zero_()
        f = lib.qpx_bench if which < 40 else lib.qpx_bench_ptr
        assert f(which, blocks, 100, 0, out.data_ptr(), inp.data_ptr(), None) == 0
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        a = o[4096:4096 + blocks]
        extra = ""
        if which == 41:
            extra = "   (the second chain: %.0f)" % o[12288:12288 + blocks].mean()
        print("  %-58s ticks per pivot block: mean %8.0f  min %8.0f  max %8.0f%s" % (names[which], a.mean(), a.min(), a.max(), extra))
