#!/usr/bin/env python3
"""A/B timing of two builds of the library ON THE SAME BOX (boxes of the pool differ by ~6 % in clocks):
ab_bench.py <libA.so>[:variant] <libB.so>[:variant] ... [B n m q]  -- fwd+bwd ms per step, the loop kernel's ms and the pre-factorisation's (KKTFactors.build),
alternating; `:variant` = the value qpx_set_ipm_variant gets for that entry (default $QPX_VARIANT or 0)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402
from qpth_amd.qp import QPFunction  # noqa: E402

entries = [x for x in sys.argv[1:] if '.so' in x]
dims = [x for x in sys.argv[1:] if '.so' not in x]
libs = [x.split(':')[0] for x in entries]
variants = [int(x.split(':')[1]) if ':' in x else int(os.environ.get("QPX_VARIANT", "0")) for x in entries]
B, n, m, q = [int(x) for x in (dims if len(dims) == 4 else (512, 100, 100, 0))]
dev = torch.device("cuda:0")
Q, p, G, h, A, b = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 0)]
p.requires_grad_(True)
ones = torch.ones(B, n, dtype=Q.dtype, device=dev)
handles = [_lib.QpxLib(os.path.abspath(x), strict=False) for x in libs]
for rep in range(3):
    for name, lib, variant in zip(entries, handles, variants):
        lib.dll.qpx_set_ipm_variant(variant)
        _lib.set_test_backend(lib)
        qpf = QPFunction(verbose=-1)
        for _ in range(5):
            z = qpf(Q, p, G, h, A, b); z.backward(ones); p.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            z = qpf(Q, p, G, h, A, b); z.backward(ones); p.grad = None
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 30 * 1e3
        fac = KKTFactors.build(Q, G, A, B)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fac.ipm(p.detach(), h, b)
        e1.record(); torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for _ in range(20):
            KKTFactors.build(Q, G, A, B)
        f1.record(); torch.cuda.synchronize()
        print("%-40s step %.4f ms   loop kernel %.4f ms   pre-factorisation %.4f ms" %
              (os.path.basename(name), step, e0.elapsed_time(e1) / 20, f0.elapsed_time(f1) / 20))
        lib.dll.qpx_set_ipm_variant(0)
        _lib.set_test_backend(None)
