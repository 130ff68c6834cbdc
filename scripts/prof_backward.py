#!/usr/bin/env python3
"""Phase breakdown of the backward kernel (kkt_mat_role, qpx_grid.h) from the profiling build (libqpx_hip_prof.so): clock
cycles thread 0 of each workgroup (chain-wave form: the chain wave) spent per phase, averaged over the batch.
    prof_backward.py [B n m q]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import problems  # noqa: E402
from csrc_layout import prof_offset  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

NAMES = ["vectors in (lam, s, dl_dz)", "products M rx, K rx", "load R, + diag", "factorisation", "solve", "M^T dz (+ N, S11 terms)", "outputs"]


def main():
    B, n, m, q = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (512, 100, 100, 0))]
    dev = torch.device("cuda:0")
    lib = _lib.QpxLib(os.path.join(ROOT, "qpth_amd", "libqpx_hip_prof.so"), strict=False)
    _lib.set_test_backend(lib)
    tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 0)]
    fac = KKTFactors.build(tQ, tG, tA, B)
    res = fac.ipm(tp, th, tb)
    ones = torch.ones(B, n, dtype=tQ.dtype, device=dev)
    want = (False, True, False, False, False, False)
    for rep in range(3):
        fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=want)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(20):
        fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=want)
    e1.record(); torch.cuda.synchronize()
    cyc = fac.blob.reshape(B, -1)[:, prof_offset(n, m, q, 1):][:, :8].double().cpu().numpy()
    print("B=%d n=%d m=%d q=%d: backward (dp only) %.4f ms (profiling build); cycles per QP (thread 0), mean %.0f max %.0f" % (
        B, n, m, q, e0.elapsed_time(e1) / 20, cyc[:, :7].sum(1).mean(), cyc[:, :7].sum(1).max()))
    for i, nm in enumerate(NAMES):
        print("  %-28s %9.0f cycles (%5.1f%%)" % (nm, cyc[:, i].mean(), 100 * cyc[:, i].sum() / cyc[:, :7].sum()))


if __name__ == "__main__":
    main()
