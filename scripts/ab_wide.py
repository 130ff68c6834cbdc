#!/usr/bin/env python3
"""Same-box A/B of the two ways float32 tensors reach the float64 kernels (C2 by default):
  host : x.double() on every parameter, QPFunction in float64, .float() on results and gradient (round-2 commit 3bd2c5e)
  wide : QPFunction(refine=None) on the float32 tensors = QPX_F32_WIDE, the kernels widen on load / narrow on store
  f64  : QPFunction on float64 tensors (the headline path), for scale
ab_wide.py [B n m q]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import problems  # noqa: E402
from qpth_amd.qp import QPFunction  # noqa: E402

dims = sys.argv[1:]
B, n, m, q = [int(x) for x in (dims if len(dims) == 4 else (512, 100, 100, 0))]
dev = torch.device("cuda:0")
t32 = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 0, np.float32)]
t64 = [x.double() for x in t32]
t32[1].requires_grad_(True)
t64[1].requires_grad_(True)
ones32 = torch.ones(B, n, dtype=torch.float32, device=dev)
ones64 = ones32.double()
qpf = QPFunction(verbose=-1)


def step_wide():
    z = qpf(*t32); z.backward(ones32); t32[1].grad = None


def step_host():
    w = [x.detach().double() for x in t32]
    w[1].requires_grad_(True)
    z = qpf(*w).float()
    z.backward(ones32)
    return w[1].grad.float()


def step_f64():
    z = qpf(*t64); z.backward(ones64); t64[1].grad = None


for rep in range(3):
    for name, fn in (("host-side casts", step_host), ("QPX_F32_WIDE (in-kernel)", step_wide), ("float64 tensors", step_f64)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        print("%-28s step %.4f ms  %.0f QPs/s" % (name, (time.perf_counter() - t0) / 100 * 1e3, B * 100 / (time.perf_counter() - t0)))
