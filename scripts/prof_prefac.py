#!/usr/bin/env python3
"""Phase breakdown of k_prefac_tile (qpx_prefac.h) from the profiling build (libqpx_hip_prof.so, -DQPX_PROFILE): clock
ticks thread 0 of each workgroup (wave 0: the chain wave of the factorisation, the owner of the last m-block) spent per
phase, averaged over the batch, + the kernel's time by events.  prof_prefac.py [B n m]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402
from csrc_layout import prof_offset  # noqa: E402

NAMES = ["Q -> LDS", "tiles out of LDS", "ldl_inv", "V -> LDS", "K tiles (share)", "blocks A, B: Yt, col sums, M^T", "barrier, stage Yt, || G^T 1 ||",
         "R tiles"]


def main():
    B, n, m = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 100, 100))]
    dev = torch.device("cuda:0")
    lib = _lib.QpxLib(os.path.join(ROOT, "qpth_amd", "libqpx_hip_prof.so"), strict=False)
    _lib.set_test_backend(lib)
    lib.dll.qpx_set_ipm_variant(int(os.environ.get("QPX_VARIANT", "0")))
    tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, 0, 0)]
    for _ in range(3):
        fac = KKTFactors.build(tQ, tG, tA, B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fac = KKTFactors.build(tQ, tG, tA, B)
    e1.record(); torch.cuda.synchronize()
    pre = fac.blob.reshape(B, -1)[:, prof_offset(n, m, 0, 1):][:, :8].double().cpu().numpy()
    print("B=%d n=%d m=%d: pre-factorisation %.4f ms (profiling build); ticks per QP (thread 0), mean %.0f" % (
        B, n, m, e0.elapsed_time(e1) / 20, pre[:, :8].sum(1).mean()))
    for i, nm in enumerate(NAMES):
        print("  %-28s %10.0f ticks (%5.1f%%)  max %10.0f" % (nm, pre[:, i].mean(), 100 * pre[:, i].sum() / pre[:, :8].sum(), pre[:, i].max()))


if __name__ == "__main__":
    main()
