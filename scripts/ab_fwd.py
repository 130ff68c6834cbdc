#!/usr/bin/env python3
"""Round-5 A/B of the forward at the C ABI (HIP events on the launch stream, same box, alternating):
   two launches (qpx_pre_factor + qpx_ipm) against qpx_forward as ONE launch (QPX_TUNE_FUSED_FORWARD = 2), each with the
   second workgroup of a CU started 0 .. N x ~8 k cycles late (QPX_TUNE_DEPHASE).
ab_fwd.py [B n m q] [dephase values ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402

args = [int(x) for x in sys.argv[1:]]
B, n, m, q = args[:4] if len(args) >= 4 else (512, 100, 100, 0)
dephases = args[4:] if len(args) > 4 else [0, 1, 2, 3, 4, 6, 8]
dev = torch.device("cuda:0")
lib = _lib.hip()
Q, p, G, h, A, b = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 0)]
elems = lib.factor_elems(_lib.QPX_F64, n, m, q)
blob = torch.empty(B * elems, dtype=torch.float64, device=dev)
status = torch.zeros(B, dtype=torch.int32, device=dev)
zhat = torch.empty(B, n, dtype=torch.float64, device=dev)
lam = torch.empty(B, m, dtype=torch.float64, device=dev)
slack = torch.empty(B, m, dtype=torch.float64, device=dev)
iters = torch.empty(B, dtype=torch.int32, device=dev)
bres = torch.empty(B, dtype=torch.float64, device=dev)


def two():
    lib.pre_factor(B, n, m, q, Q, G, None, blob, status)
    lib.ipm(B, n, m, q, p, h, None, blob, elems, 1e-12, 20, 3, _lib.STALL_FLOOR, zhat, None, lam, slack, iters, status, bres)


def one():
    lib.forward(B, n, m, q, Q, p, G, h, None, None, blob, 1e-12, 20, 3, _lib.STALL_FLOOR, zhat, None, lam, slack, iters, status, bres)


def timed(fn, nrep=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(nrep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep


lib.dll.qpx_set_tuning(_lib.TUNE_FUSED_FORWARD, 1)
lib.dll.qpx_set_tuning(_lib.TUNE_DEPHASE, 0)
two()
torch.cuda.synchronize()
z_ref = zhat.clone()
it_ref = iters.clone()
print("B=%d n=%d m=%d q=%d   iterations mean %.2f max %d" % (B, n, m, q, it_ref.float().mean().item(), int(it_ref.max())))
for rep in range(3):
    for dp in dephases:
        lib.dll.qpx_set_tuning(_lib.TUNE_DEPHASE, dp)
        lib.dll.qpx_set_tuning(_lib.TUNE_FUSED_FORWARD, 1)
        t2 = timed(two)
        lib.dll.qpx_set_tuning(_lib.TUNE_FUSED_FORWARD, 2)
        zhat.zero_()
        t1 = timed(one)
        err = float((zhat - z_ref).abs().max())
        same_it = bool((iters == it_ref).all())
        print("dephase %2d   two launches %.4f ms   one launch %.4f ms   (one launch: max |dz| vs two %.1e, iterations equal: %s)"
              % (dp, t2, t1, err, same_it))
lib.dll.qpx_set_tuning(_lib.TUNE_DEPHASE, 0)
lib.dll.qpx_set_tuning(_lib.TUNE_FUSED_FORWARD, 0)
