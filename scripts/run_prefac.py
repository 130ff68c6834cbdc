#!/usr/bin/env python3
"""KKTFactors.build in a loop (for rocprofv3): run_prefac.py [B n m reps]; QPX_VARIANT selects the form."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

B, n, m, reps = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (512, 100, 100, 10))]
_lib.hip().dll.qpx_set_ipm_variant(int(os.environ.get("QPX_VARIANT", "0")))
dev = torch.device("cuda:0")
tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, 0, 0)]
for _ in range(reps):
    fac = KKTFactors.build(tQ, tG, tA, B)
torch.cuda.synchronize()
print("done", fac.blob.shape)
