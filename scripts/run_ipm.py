#!/usr/bin/env python3
"""The loop kernel alone in a loop (for rocprofv3 counter passes): run_ipm.py [B n m q reps]; QPX_VARIANT selects the form.
Inputs generated on the device (prof-linear.py's generator in torch)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_batch  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

B, n, m, q, reps = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (65536, 64, 64, 0, 3))]
_lib.hip().dll.qpx_set_ipm_variant(int(os.environ.get("QPX_VARIANT", "0")))
dev = torch.device("cuda:0")
Q, p, G, h, A, b = device_batch(B, n, m, dev, torch.float64, 7)
fac = KKTFactors.build(Q, G, A, B)
for _ in range(reps):
    res = fac.ipm(p, h, b)
torch.cuda.synchronize()
print("done", float(res.iters.float().mean()))
