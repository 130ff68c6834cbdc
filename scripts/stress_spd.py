#!/usr/bin/env python3
"""Repeated pre-factorisation of a large batch of SPD matrices: any QP flagged not-SPD is a bug
(rare-race hunt).  Prints the flagged indices per repetition."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

dev = torch.device("cuda:0")
for (B, n, m) in ((4096, 64, 64), (2048, 100, 100), (8192, 32, 32)):
    arrs = problems.prof_qp(B, n, m, 0, 0)
    tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in arrs]
    for rep in range(6):
        fac = KKTFactors.build(tQ, tG, tA, B)
        torch.cuda.synchronize()
        st = fac.status.cpu().numpy()
        bad = np.nonzero(st)[0]
        res = fac.ipm(tp, th, tb)
        torch.cuda.synchronize()
        st2 = res.status.cpu().numpy()
        print("B=%d n=%d rep %d: prefactor flagged %s ; after ipm nonzero status at %s values %s" % (
            B, n, rep, bad.tolist()[:8], np.nonzero(st2)[0].tolist()[:8], sorted(set(st2.tolist()))))
