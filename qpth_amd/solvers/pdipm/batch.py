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


def _diag_of(D):
    """the reference passes D = diag(d) as a (nBatch, nineq, nineq) matrix to its full-system solvers"""
    return torch.diagonal(D, dim1=-2, dim2=-1).contiguous() if D.dim() >= 2 and D.size(-1) == D.size(-2) and D.dim() == 3 else D


def solve_kkt_ir(*args, niter=1):
    """solve_kkt_ir (batch.py:244-270): solve, then `niter` steps of iterative refinement on the residual of the
    original KKT system (kkt_resid_reg, batch.py:228-241) -- inside the kernel, with the factorisation re-used.
    Two call forms:
      solve_kkt_ir(Q, D, G, A, rx, rs, rz, ry, niter=1)             the reference's (D = diag(d) as a matrix): factors first
      solve_kkt_ir(Q_LU, d, G, A, S_LU, rx, rs, rz, ry, niter=1)    on handles from pre_factor_kkt
    (No eps-regularisation: the un-pivoted symmetric factorisations here need none, and the correction is added with
    the right sign -- DESIGN.md, deliberate differences 7.)  Raises where the kernel family that serves the size has no
    in-kernel refinement (nz + neq + nineq > 208: qpx_refine_supported)."""
    if isinstance(args[0], _Handle):
        Q_LU, d, G, A, S_LU, rx, rs, rz, ry = args[:9]
        if len(args) > 9:
            niter = args[9]
        return Q_LU.fac.solve_kkt(d, rx, rs, rz, ry, refine=niter)
    Q, D, G, A, rx, rs, rz, ry = args[:8]
    if len(args) > 8:
        niter = args[8]
    fac = _dp.KKTFactors.build(Q, G, A)
    fac.raise_on_failure()
    return fac.solve_kkt(_diag_of(D), rx, rs, rz, ry, refine=niter)


def factor_solve_kkt(Q, D, G, A, rx, rs, rz, ry):
    """factor_solve_kkt (batch.py:313-346, KKTSolvers.LU_FULL): the whole KKT system factored and solved in one call,
    D = diag(d) as a (nBatch, nineq, nineq) matrix.  The reference eliminates (x, s) first, this library z first --
    two orders of one system (test.py:222-234 checks they agree); here: pre_factor_kkt + the fused factor / solve."""
    fac = _dp.KKTFactors.build(Q, G, A)
    fac.raise_on_failure()
    return fac.solve_kkt(_diag_of(D), rx, rs, rz, ry)


def factor_solve_kkt_reg(Q_tilde, D, G, A, rx, rs, rz, ry, eps):
    """factor_solve_kkt_reg (batch.py:273-310): the KKT system with -eps I in the (z, z) and (y, y) blocks,
        Q~ dx + G^T dz + A^T dy = -rx,  D ds + dz = -rs,  G dx + ds - eps dz = -rz,  A dx - eps dy = -ry.
    The (z, z) part is a change of the diagonal -- eliminating ds leaves G dx - (1/d + eps) dz = -rz + rs/d, i.e. the
    un-regularised system with d' = d / (1 + eps d) and rs' = rs d'/d -- and runs on the same kernels.  The (y, y) part
    (round 4; refused until then) is a rank-neq correction of that system: its last row reads A dx = -(ry - eps dy), and
    the solution is linear in ry, so with Y = d(dy)/d(ry) (neq solves with unit right-hand sides, independent of the
    caller's) dy solves (I + eps Y) dy = dy0 and one more solve with ry - eps dy gives the rest: neq + 2 launches of the
    fused factor / solve kernel and one neq x neq system per QP (qpx_dense_solve)."""
    nineq, nz, neq, nBatch = get_sizes(G, A)
    d = _diag_of(D)
    fac = _dp.KKTFactors.build(Q_tilde, G, A)
    fac.raise_on_failure()
    dreg = d / (1.0 + eps * d)
    rs_ = rs if rs is not None else torch.zeros_like(rz if rz is not None else d.expand(nBatch, nineq))
    rs_reg = rs_ / (1.0 + eps * d)
    ry_eff = ry
    if neq > 0:
        dreg_b = dreg if dreg.dim() == 2 else dreg.unsqueeze(0).expand(nBatch, nineq)
        ry0 = ry if ry is not None else torch.zeros(nBatch, neq, dtype=Q_tilde.dtype, device=Q_tilde.device)
        dy0 = fac.solve_kkt(dreg_b, rx, rs_reg, rz, ry0)[3]
        eye = torch.eye(neq, dtype=Q_tilde.dtype, device=Q_tilde.device)
        cols = [fac.solve_kkt(dreg_b, None, None, None, eye[j].expand(nBatch, neq).contiguous())[3] for j in range(neq)]
        Y = torch.stack(cols, dim=2)                                     # Y[:, :, j] = dy for ry = e_j
        # (I + eps Y) dy = dy0: one general neq x neq system per QP, by the library's pivoted elimination kernel
        st = torch.zeros(nBatch, dtype=torch.int32, device=Q_tilde.device)
        dy_reg = fac.lib.dense_solve((eye + eps * Y).contiguous(), dy0.clone().contiguous(), st)
        # the reference's torch.linalg.solve raised on a singular system (batch.py:294-303 is an LU solve that fails);
        # the kernel reports it per QP and returns NaNs -- read the words (this entry point is off the QPFunction path: the
        # one small D2H copy and the wait are what torch's own error check costs too) and raise as the reference does
        if int(st.max().item()) & _dp._lib.ST_KKT_BREAKDOWN:
            raise RuntimeError("qpth_amd: factor_solve_kkt_reg: the regularised (y, y) block (I + eps Y) is singular for QP(s) %s"
                               % torch.nonzero(st).flatten().tolist()[:8])
        ry_eff = ry0 - eps * dy_reg
    dx, _, dz, dy = fac.solve_kkt(dreg, rx, rs_reg, rz, ry_eff)
    ds = (-rs_ - dz) / (d if d.dim() == 2 else d.unsqueeze(0))      # the second block row with the caller's d
    return dx, ds, dz, dy

def unpack_kkt(v, nz, nineq, neq):
    """unpack_kkt (batch.py:216-225): the stacked KKT vector (x, s, z, y) -> its four parts (views)."""
    x = v[:, :nz]
    s = v[:, nz:nz + nineq]
    z = v[:, nz + nineq:nz + 2 * nineq]
    y = v[:, nz + 2 * nineq:nz + 2 * nineq + neq]
    return x, s, z, y


def kkt_resid_reg(Q_tilde, D_tilde, G, A, eps, dx, ds, dz, dy, rx, rs, rz, ry):
    """kkt_resid_reg (batch.py:228-241): the residual of (dx, ds, dz, dy) in the regularised KKT system
        Q~ dx + G^T dz + A^T dy + rx,   D~ ds + dz + rs,   G dx + ds - eps dz + rz,   A dx - eps dy + ry,
    D_tilde a (nBatch, nineq, nineq) matrix as in the reference, dy / ry None without equality constraints.  A DIAGNOSTIC
    helper kept for callers of the reference's module surface: nothing on the product path calls it -- the kernels that
    refine (qpx_factor_solve_kkt(..., refine), qpx_polish) form this residual themselves, in float64 accumulation -- and it
    is the one function of this file that is plain tensor arithmetic (broadcast products, no library calls)."""
    def mv(M, v):            # (B, r, c) x (B, c) -> (B, r)
        return (M * v.unsqueeze(1)).sum(2)

    def mtv(M, v):           # (B, r, c)^T x (B, r) -> (B, c)
        return (M * v.unsqueeze(2)).sum(1)

    resx = mv(Q_tilde, dx) + mtv(G, dz) + rx
    if dy is not None:
        resx = resx + mtv(A, dy)
    ress = mv(D_tilde, ds) + dz + rs
    resz = mv(G, dx) + ds - eps * dz + rz
    resy = mv(A, dx) - eps * dy + ry if dy is not None else None
    return resx, ress, resz, resy


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
        res = fac.polish(p, h, b, res, refine=1)          # (its solves refined as well: solve_kkt_ir is what the name asks for)
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
