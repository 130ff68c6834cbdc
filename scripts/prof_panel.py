#!/usr/bin/env python3
"""Sub-phase breakdown of one panel of the tile kernel's blocked LDL^T (libqpx_hip_pprof.so,
`make -C qpth_amd/csrc panelprof`): shader-clock ticks thread 0 of each QP spent in publish /
barrier / pivot block / operands / update, per factorisation and per panel.  The timers force
`s_waitcnt 0` at every cut, so overlap across cuts is lost -- read the split, not the sum."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

# sixteen-column panels (the shipped build); the four-column form (-DQPX_TILE_PANEL4) cuts at other places
NAMES16 = ["publish (X, S)", "pivot block (when wave 0 owns it)", "barrier A (waits for the pivot block)", "operand tiles (LDS + 4 mfma per J)", "barrier B", "update of the previous panel (LDS + mfma)"]
NAMES4 = ["publish", "barrier", "S read + 4x4 + masks", "operands (LDS + fma)", "assign + mfma issue", "entry (mfma drain)"]
PANEL = int(os.environ.get("QPX_PANEL_COLS", "16"))
NAMES = NAMES16 if PANEL == 16 else NAMES4


def main():
    B, n, m, q = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (512, 100, 100, 0))]
    dev = torch.device("cuda:0")
    lib = _lib.QpxLib(os.path.join(ROOT, "qpth_amd", "libqpx_hip_pprof.so"), strict=False)
    _lib.set_test_backend(lib)
    lib.dll.qpx_set_ipm_variant(int(os.environ.get("QPX_VARIANT", "0")))
    arrs = problems.prof_qp(B, n, m, q, 0, np.float64)
    tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in arrs]
    fac = KKTFactors.build(tQ, tG, tA, B)
    out = (ctypes.c_ulonglong * 8)()
    for rep in range(2):
        if hasattr(lib.dll, "qpx_chain_prof_read"):
            lib.dll.qpx_chain_prof_read((ctypes.c_ulonglong * 20)())     # reset
        res = fac.ipm(tp, th, tb)
        torch.cuda.synchronize()
        lib.dll.qpx_panel_prof_read(out)
    if hasattr(lib.dll, "qpx_chain_prof_read"):
        co = (ctypes.c_ulonglong * 20)()
        lib.dll.qpx_chain_prof_read(co)
        cc = np.array(list(co), dtype=np.float64)
        if cc[16] > 0:
            npan = (m + 15) // 16
            print("B=%d n=%d m=%d q=%d chain-wave form: %d factorisations; ticks per PANEL" % (B, n, m, q, cc[16]))
            print("  %-12s %12s %12s %12s %12s %10s" % ("wave", "interval 1", "wait at Y", "interval 2", "wait at X", "sum"))
            for w, nm in enumerate(["chain", "tile 0", "tile 1", "tile 2"]):
                v = cc[4 * w:4 * w + 4] / cc[16] / npan
                print("  %-12s %12.0f %12.0f %12.0f %12.0f %10.0f" % (nm, v[0], v[1], v[2], v[3], v.sum()))
            return
    c = np.array(list(out), dtype=np.float64)
    nfac = c[6]
    npan = (m + PANEL - 1) // PANEL
    print("B=%d n=%d m=%d q=%d variant=%s: %d factorisations, %d panels each; ticks per factorisation %.0f" % (
        B, n, m, q, os.environ.get("QPX_VARIANT", "0"), nfac, npan, c[:6].sum() / nfac))
    for i, nm in enumerate(NAMES):
        print("  %-24s %10.0f per factorisation  %8.0f per panel (%5.1f%%)" % (nm, c[i] / nfac, c[i] / nfac / npan, 100 * c[i] / c[:6].sum()))


if __name__ == "__main__":
    main()
