#!/usr/bin/env python3
"""Where the finishing kernel (qpx_polish) spends its time: launches with (steps, refine) in {(0,0), (1,0), (1,1), (2,1)}
timed with HIP events on the launch stream, float32 and float64, at B n m q (default C2) -- steps = 0 is the fixed
cost (load the iterate, one residual evaluation), refine = 0 -> 1 the cost of the two refinement passes of a step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

dims = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else [512, 100, 100, 0]
B, n, m, q = dims
dev = torch.device("cuda:0")
for dt, npd in ((torch.float32, np.float32), (torch.float64, np.float64)):
    Q, p, G, h, A, b = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 0, npd)]
    fac = KKTFactors.build(Q, G, A, B)
    res0 = fac.ipm(p, h, b)
    torch.cuda.synchronize()
    row = []
    for steps, refine in ((0, 0), (1, 0), (1, 1), (2, 1)):
        def run():
            res = type(res0)()
            for k in res0.__slots__:
                v = getattr(res0, k)
                setattr(res, k, v.clone() if torch.is_tensor(v) else v)
            return fac.polish(p, h, b, res, steps=steps, refine=refine)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        row.append("steps=%d refine=%d: %.3f ms" % (steps, refine, e0.elapsed_time(e1) / 20))
    print("B=%d n=%d m=%d q=%d %s (incl. ~5 small clones per call): %s" % (B, n, m, q, str(dt).replace("torch.", ""), " | ".join(row)))
