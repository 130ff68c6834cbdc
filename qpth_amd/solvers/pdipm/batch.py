"""MI355X drop-in for qpth/solvers/pdipm/batch.py -- same four solver entry points, same
argument meaning and return order, executed by the HIP kernels behind include/qpx.h:

    pre_factor_kkt(Q, G, A) -> (Q_LU, S_LU, R)                         batch.py:375-429
    factor_kkt(S_LU, R, d)                  (in place)                 batch.py:435-470
    solve_kkt(Q_LU, d, G, A, S_LU, rx, rs, rz, ry) -> dx, ds, dz, dy   batch.py:349-372
    forward(Q, p, G, h, A, b, Q_LU, S_LU, R, eps, verbose,
            notImprovedLim, maxIter, solver) -> x, y, z, s             batch.py:47-207

Differences that are visible to a caller, all documented in DESIGN.md:
  * Q_LU / S_LU / R are opaque handles onto one device-resident factor blob (Cholesky based;
    the reference's GPU branch of lu_hack is un-pivoted as well, batch.py:8-20) -- they can be
    passed around exactly like the reference's tuples but not indexed;
  * factor_kkt only checks `d`: the m x m factor lives in registers / LDS and is built by the kernel
    that consumes it (solve_kkt / backward), which is cheaper than a round trip through HBM;
  * the IPM loop runs per QP (one workgroup each); the reference's batch-global stopping
    test and get_step quirk are replaced by their batch-of-one meaning (`stall_policy`).
"""
from enum import Enum

import torch

from ... import kkt as _dp
from ..._lib import STALL_FLOOR, STALL_OFF, STALL_REFERENCE  # noqa: F401
from ...util import get_sizes

INACC_ERR = """
--------
qpth warning: Returning an inaccurate and potentially incorrect solution.

Some residual is large.
Your problem may be infeasible or difficult.

You can try using the CVXPY solver to see if your problem is feasible
and you can use the verbose option to check the convergence status of
our solver while increasing the number of iterations.

Advanced users:
You can also try to enable iterative refinement in the solver:
https://github.com/locuslab/qpth/issues/6
--------
"""


class KKTSolvers(Enum):
    LU_FULL = 1
    LU_PARTIAL = 2
    IR_UNOPT = 3


class _Handle:
    """What pre_factor_kkt returns three times over (as Q_LU, S_LU and R): a view onto the
    shared KKTFactors of the batch."""

    def __init__(self, fac, role):
        self.fac, self.role = fac, role

    def __repr__(self):
        return "<qpth_amd %s handle: B=%d nz=%d nineq=%d neq=%d %s>" % (
            self.role, self.fac.B, self.fac.n, self.fac.m, self.fac.q, self.fac.blob.dtype)


def pre_factor_kkt(Q, G, A):
    """Perform all one-time factorizations and cache relevant matrix products."""
    fac = _dp.KKTFactors.build(Q, G, A)
    fac.raise_on_failure()
    return _Handle(fac, "Q_LU"), _Handle(fac, "S_LU"), _Handle(fac, "R")


def factor_kkt(S_LU, R, d):
    """Factor the U22 block that we can only do after we know D.

    Here the m x m factor of R + diag(1/d) never leaves the registers / LDS of the kernel that consumes it
    (solve_kkt below takes `d` again, exactly as the reference's signature does), so this call only checks
    its arguments."""
    if d.size(-1) != S_LU.fac.m:
        raise RuntimeError("factor_kkt: d has %d entries, nineq is %d" % (d.size(-1), S_LU.fac.m))


def solve_kkt(Q_LU, d, G, A, S_LU, rx, rs, rz, ry):
    """Solve KKT equations for the affine step."""
    fac = Q_LU.fac
    return fac.solve_kkt(d, rx, rs, rz, ry)


def solve_kkt_ir(Q_LU, d, G, A, S_LU, rx, rs, rz, ry, niter=1):
    """The reference's solve_kkt_ir (batch.py:244-270) on the pre-factored handles: solve, then `niter` steps of
    iterative refinement on the residual of the original KKT system (kkt_resid_reg, batch.py:228-241) -- inside the
    kernel, with the factorisation re-used.  (No eps-regularisation: the un-pivoted factorisations here need none.)"""
    return Q_LU.fac.solve_kkt(d, rx, rs, rz, ry, refine=niter)


def forward(Q, p, G, h, A, b, Q_LU, S_LU, R, eps=1e-12, verbose=0, notImprovedLim=3,
            maxIter=20, solver=KKTSolvers.LU_PARTIAL, stall_policy=None):
    """
    Q_LU, S_LU, R = pre_factor_kkt(Q, G, A)
    """
    if not isinstance(solver, KKTSolvers):
        raise ValueError("solver must be a KKTSolvers member, got %r" % (solver,))
    nineq, nz, neq, nBatch = get_sizes(G, A)
    fac = Q_LU.fac
    # LU_FULL and LU_PARTIAL are two elimination orders of one KKT system in the reference (batch.py:313-346 vs
    # 349-372, same iterates to rounding: test.py:222-234); the HIP path has one elimination, the condensed one,
    # and runs it for both.  IR_UNOPT adds what its name promises: steps on the residual of the ORIGINAL system.
    res = fac.ipm(p, h, b, eps, maxIter, notImprovedLim, stall_policy, want_trace=(verbose == 1))
    if solver == KKTSolvers.IR_UNOPT:
        res = fac.polish(p, h, b, res)
    if verbose == 1:
        tr = res.trace.cpu()
        it_max = int(res.iters.max().item())
        for i in range(it_max):
            act = (res.iters > i).cpu()
            row = tr[i][act]
            print('iter: {}, pri_resid: {:.5e}, dual_resid: {:.5e}, mu: {:.5e}'.format(
                i, row[:, 0].mean(), row[:, 1].mean(), row[:, 2].mean()))
    if verbose >= 0:
        # batch.py:141-142,205-206: print (not raise) when some best residual is > 1
        if bool((res.best_resid > 1.).any().item()) or not bool(torch.isfinite(res.best_resid).all().item()):
            print(INACC_ERR)
    return res.zhat, (res.nu if neq > 0 else None), res.lam, res.slacks
