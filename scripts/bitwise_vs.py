#!/usr/bin/env python3
"""Are two builds of the library bit-identical in their results?  bitwise_vs.py a.so b.so [B n m q]: forward (zhat, lam, slacks, nu,
iters) and the p-gradient of both on the same seeded inputs, compared with torch.equal."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

libs = [x for x in sys.argv[1:] if ".so" in x]
dims = [int(x) for x in sys.argv[1:] if ".so" not in x] or [512, 100, 100, 0]
B, n, m, q = dims
dev = torch.device("cuda:0")
Q, p, G, h, A, b = [torch.tensor(x, device=dev) for x in problems.prof_qp(B, n, m, q, 5)]
ones = torch.ones(B, n, dtype=Q.dtype, device=dev)
outs = []
for path in libs:
    lib = _lib.QpxLib(os.path.abspath(path), strict=False)
    _lib.set_test_backend(lib)
    fac = KKTFactors.build(Q, G, A, B)
    res = fac.ipm(p, h, b)
    grads = fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=(False, True, False, False, False, False))
    torch.cuda.synchronize()
    outs.append([res.zhat.clone(), res.lam.clone(), res.slacks.clone(), res.nu.clone(), res.iters.clone(), grads[1].clone()])
    _lib.set_test_backend(None)
names = ["zhat", "lam", "slacks", "nu", "iters", "dp"]
same = [torch.equal(x, y) for x, y in zip(outs[0], outs[1])]
print("B=%d n=%d m=%d q=%d  %s vs %s: %s" % (B, n, m, q, os.path.basename(libs[0]), os.path.basename(libs[1]),
      "BIT-IDENTICAL" if all(same) else "differ in " + ", ".join(nm for nm, s in zip(names, same) if not s)))
if not all(same):
    it0, it1 = outs[0][4], outs[1][4]
    zr = ((outs[0][0] - outs[1][0]).norm(dim=1) / outs[0][0].norm(dim=1)).cpu()
    print("   QPs whose iteration count differs: %d of %d (by at most %d); zhat relative difference per QP: median %.2e, max %.2e; among QPs with equal counts: max %.2e" % (
        int((it0 != it1).sum()), B, int((it0 - it1).abs().max()), zr.median().item(), zr.max().item(),
        zr[(it0 == it1).cpu()].max().item() if bool((it0 == it1).any()) else float("nan")))
    for nm, x, y in zip(names, outs[0], outs[1]):
        if x.dtype.is_floating_point and x.numel():
            print("   %-7s max abs diff %.3e (max abs %.3e)" % (nm, (x - y).abs().max().item(), x.abs().max().item()))
