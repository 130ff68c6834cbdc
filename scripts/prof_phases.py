#!/usr/bin/env python3
"""Phase breakdown of k_ipm from the profiling build (libqpx_hip_prof.so, -DQPX_PROFILE):
shader-clock cycles thread 0 of each workgroup spent per phase, averaged over the batch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

NAMES = ["consts(p,h,b)", "vector work (w0)", "copy R->T", "symv R z'", "residuals(w0)", "factorisation", "solves", "epilogue"]


def main():
    B, n, m, q = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (512, 100, 100, 0))]
    dt = np.float32 if (len(sys.argv) > 5 and sys.argv[5] == "f32") else np.float64
    dev = torch.device("cuda:0")
    lib = _lib.QpxLib(os.path.join(ROOT, "qpth_amd", "libqpx_hip_prof.so"), strict=False)
    _lib.set_test_backend(lib)        # explicit: route this script's calls to the profiling build
    lib.dll.qpx_set_ipm_variant(int(os.environ.get("QPX_VARIANT", "0")))
    arrs = problems.prof_qp(B, n, m, q, 0, dt)
    tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in arrs]
    fac = KKTFactors.build(tQ, tG, tA, B)
    for rep in range(3):
        res = fac.ipm(tp, th, tb, want_trace=True)
        torch.cuda.synchronize()
    from csrc_layout import prof_offset
    images = 0 if (int(os.environ.get("QPX_VARIANT", "0")) & 255) == 1 or n + m + q > 208 else 1
    pre = fac.blob.reshape(B, -1)[:, prof_offset(n, m, q, images):][:, :8].double().cpu().numpy()
    pn = ["load Q + chol(Q)", "load G^T, |G^T 1|", "TRSM Z=L^-1 G^T", "equality block", "SYRK R=Z^T Z", "r1 + spill to blob", "-", "-"]
    if lib.dll.qpx_set_ipm_variant(int(os.environ.get("QPX_VARIANT", "0"))) is None:
        pass
    if int(os.environ.get("QPX_VARIANT", "0")) in (1, 2):
        print("k_prefactor: total cycles/QP mean %.0f" % pre.sum(1).mean())
        for i in range(6):
            print("  %-20s %12.0f cycles (%5.1f%%)" % (pn[i], pre[:, i].mean(), 100 * pre[:, i].sum() / pre.sum()))
    else:
        sn = ["load Q, G, A", "|| G^T 1 ||", "sweep n+q pivots", "scatter to the blob"]
        print("k_sweep: total cycles/QP mean %.0f" % pre[:, :4].sum(1).mean())
        for i in range(4):
            print("  %-20s %12.0f cycles (%5.1f%%)" % (sn[i], pre[:, i].mean(), 100 * pre[:, i].sum() / pre[:, :4].sum()))
    cyc = res.trace.reshape(-1)[:B * 8].reshape(B, 8).double().cpu().numpy()
    iters = res.iters.cpu().numpy()
    tot = cyc.sum(1)
    print("B=%d n=%d m=%d q=%d %s  iterations mean %.2f  total cycles/QP mean %.0f max %.0f" % (
        B, n, m, q, dt.__name__, iters.mean(), tot.mean(), tot.max()))
    for i, nm in enumerate(NAMES):
        per_it = cyc[:, i].mean() / (iters.mean() + 1) if 1 <= i <= 6 else float("nan")
        print("  %-16s %12.0f cycles (%5.1f%%)   per iteration %10.0f" % (nm, cyc[:, i].mean(), 100 * cyc[:, i].sum() / tot.sum(), per_it))


if __name__ == "__main__":
    main()
