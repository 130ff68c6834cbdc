#!/usr/bin/env python3
"""Localise accuracy differences between kernel variants on the GPU: factor_solve_kkt residuals and
IPM residual trajectories (variant 0 = thread-grid kernels, 1 = workgroup kernels)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems
from qpth_amd import _lib
from qpth_amd.kkt import KKTFactors
from oracle import qp_oracle as orc
dev = torch.device("cuda:0")
B, n, m, q = 64, 100, 100, 0
arrs = problems.prof_qp(B, n, m, q, 0)
Q, p, G, h, A, b = arrs
tq = [torch.tensor(x, device=dev) for x in arrs]
o = orc.OracleQP(Q, p, G, h, A, b)
xr, yr, zr, sr, info = o.forward()
lib = _lib.hip()
for variant in (1, 0):
    lib.dll.qpx_set_ipm_variant(variant)
    fac = KKTFactors.build(tq[0], tq[2], tq[4], B)
    res = fac.ipm(tq[1], tq[3], tq[5], want_trace=True)
    torch.cuda.synchronize()
    err = np.linalg.norm(res.zhat.cpu().numpy() - xr, axis=1) / np.linalg.norm(xr, axis=1)
    it = res.iters.cpu().numpy()
    print("variant %d: zhat rel err max %.2e median %.2e ; iters mean %.2f ; worst QPs %s" % (variant, err.max(), np.median(err), it.mean(), np.argsort(err)[-4:].tolist()))
    w = int(np.argmax(err))
    tr = res.trace.cpu().numpy()[:, w, :]
    print("   QP %d (iters %d, best_resid %.2e) pri_resid trajectory: %s" % (w, it[w], res.best_resid[w].item(), " ".join("%.1e" % v for v in tr[:it[w], 0])))
    print("   mu trajectory: %s" % " ".join("%.1e" % v for v in tr[:it[w], 2]))
    # one KKT solve with a badly scaled d (late-IPM like) vs numpy
    rng = np.random.RandomState(0)
    dnp = np.exp(rng.uniform(-20, 20, size=(B, m)))
    rx = rng.randn(B, n); rs = rng.randn(B, m); rz = rng.randn(B, m)
    dx, ds, dz, dy = fac.solve_kkt(torch.tensor(dnp, device=dev), torch.tensor(rx, device=dev), torch.tensor(rs, device=dev), torch.tensor(rz, device=dev), None)
    torch.cuda.synchronize()
    dx, ds, dz = dx.cpu().numpy(), ds.cpu().numpy(), dz.cpu().numpy()
    # residuals of the KKT system  Q dx + G^T dz = -rx ; d ds + dz = -rs ; G dx + ds = -rz
    r1 = np.einsum("bij,bj->bi", Q, dx) + np.einsum("bmi,bm->bi", G, dz) + rx
    r2 = dnp * ds + dz + rs
    r3 = np.einsum("bmi,bi->bm", G, dx) + ds + rz
    sc = np.abs(rx).max() + np.abs(rz).max()
    print("   factor_solve_kkt residuals (max abs): stationarity %.2e  compl %.2e  primal %.2e (scale %.1f)" % (np.abs(r1).max(), np.abs(r2 / np.maximum(1, dnp)).max(), np.abs(r3).max(), sc))
