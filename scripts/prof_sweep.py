#!/usr/bin/env python3
"""Per-wave interval timers of the tile sweep (libqpx_hip_pprof.so, `make -C qpth_amd/csrc panelprof`) and its launch time."""
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

B, n, m, q = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (512, 100, 100, 0))]
dev = torch.device("cuda:0")
lib = _lib.QpxLib(os.path.join(ROOT, "qpth_amd", "libqpx_hip_pprof.so"))
_lib.set_test_backend(lib)
lib.dll.qpx_set_ipm_variant(int(os.environ.get("QPX_VARIANT", "0")))
tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 0, np.float64)]
co = (ctypes.c_ulonglong * 20)()
for rep in range(3):
    lib.dll.qpx_sweep_prof_read(co)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fac = KKTFactors.build(tQ, tG, tA, B)
    e1.record(); torch.cuda.synchronize()
lib.dll.qpx_sweep_prof_read(co)
cc = np.array(list(co), dtype=np.float64)
npan = (n + 15) // 16 + (q + 15) // 16
print("B=%d n=%d m=%d q=%d tile sweep: launch %.1f us; %d sweeps; ticks per PANEL (%d panels)" % (B, n, m, q, e0.elapsed_time(e1) * 1e3, cc[16], npan))
print("  %-12s %12s %12s %12s %12s %10s" % ("wave", "interval 1", "wait at Y", "interval 2", "wait at X", "sum"))
for w, nm in enumerate(["chain", "tile 0", "tile 1", "tile 2"]):
    v = cc[4 * w:4 * w + 4] / max(cc[16], 1) / npan
    print("  %-12s %12.0f %12.0f %12.0f %12.0f %10.0f" % (nm, v[0], v[1], v[2], v[3], v.sum()))
print("  phases per QP (thread 0): load %.0f  sweep %.0f  store %.0f ticks" % tuple(cc[17:20] / max(cc[16], 1)))
