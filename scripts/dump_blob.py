#!/usr/bin/env python3
"""Dump the factor blob and a KKT solve of a few QPs (GPU or emulator) for bitwise comparison."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems
from qpth_amd.kkt import KKTFactors
mode, out = sys.argv[1], sys.argv[2]
n, m, q = 100, 100, 0
Q, p, G, h, A, b = problems.prof_qp(64, n, m, q, 0)
sel = [6, 45]
Q, p, G, h = Q[sel], p[sel], G[sel], h[sel]
rng = np.random.RandomState(0)
dnp = np.exp(rng.uniform(-20, 20, size=(64, m)))[sel]
rx = rng.randn(64, n)[sel]; rs = rng.randn(64, m)[sel]; rz = rng.randn(64, m)[sel]
def run(dev):
    tq = [torch.tensor(x, device=dev) for x in (Q, p, G, h)]
    e = torch.empty(0, dtype=torch.float64, device=dev)
    fac = KKTFactors.build(tq[0], tq[2], e, 2)
    dx, ds, dz, dy = fac.solve_kkt(torch.tensor(dnp, device=dev), torch.tensor(rx, device=dev), torch.tensor(rs, device=dev), torch.tensor(rz, device=dev), None)
    res = fac.ipm(tq[1], tq[3], e)
    return dict(blob=fac.blob.cpu().numpy(), dx=dx.cpu().numpy(), dz=dz.cpu().numpy(), zhat=res.zhat.cpu().numpy(), iters=res.iters.cpu().numpy())
if mode == "gpu":
    r = run(torch.device("cuda:0"))
else:
    from emu.harness import emulated
    with emulated(256):
        r = run(torch.device("cpu"))
np.savez(out, **r)
print("saved", out, r["iters"])
