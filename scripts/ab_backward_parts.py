#!/usr/bin/env python3
"""Large-QP family: the backward (qpx_backward) with one part (knob bits 16..19 = 1) against the default two parts, by HIP
events and by the host's clock around a synchronised call (the step pays the larger of the two): ab_backward_parts.py B n m q"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

B, n, m, q = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 500, 500, 0))]
dev = torch.device("cuda:0")
Q, p, G, h, A, b = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 0)]
ones = torch.ones(B, n, dtype=Q.dtype, device=dev)
want = (False, True, False, False, False, False)
lib = _lib.hip()
for rep in range(2):
    for variant, name in ((65536, "one part"), (0, "two parts (default)")):
        lib.dll.qpx_set_ipm_variant(variant)
        fac = KKTFactors.build(Q, G, A, B)
        res = fac.ipm(p, h, b)
        for _ in range(3):
            fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=want)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(20):
            fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=want)
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_wall = time.perf_counter() - t0
        print("B=%d n=%d m=%d q=%d  %-20s backward: %.3f ms by events, host enqueue %.3f ms, wall %.3f ms per call" % (
            B, n, m, q, name, e0.elapsed_time(e1) / 20, t_host / 20 * 1e3, t_wall / 20 * 1e3), flush=True)
        lib.dll.qpx_set_ipm_variant(0)
