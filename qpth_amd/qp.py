"""The autograd surface of the reference, unchanged: qpth/qp.py:13-183.

    QPFunction(eps=1e-12, verbose=0, notImprovedLim=3, maxIter=20,
               solver=QPSolvers.PDIPM_BATCHED, check_Q_spd=True)(Q, p, G, h, A, b) -> zhat (nBatch, nz)

solves a batch of QPs  z* = argmin 1/2 z'Qz + p'z  s.t. Gz <= h, Az = b  and is differentiable
in all six parameters.  Any subset of the parameters may be un-batched; an empty tensor
means "no such constraint" (qp.py:58-61).  The forward runs two HIP kernels
(pre_factor_kkt, PDIPM loop), the backward one (factor_kkt + solve_kkt + gradient outer
products); state crosses from forward to backward on ctx exactly as in the reference.
"""
from enum import Enum

import torch
from torch.autograd import Function

from . import _lib
from .kkt import KKTFactors
from .solvers.pdipm import batch as pdipm_b
from .util import expandParam, extract_nBatch


class QPSolvers(Enum):
    PDIPM_BATCHED = 1
    CVXPY = 2


def _print_trace(res):
    tr = res.trace.cpu()
    iters = res.iters.cpu()
    for i in range(int(iters.max().item())):
        row = tr[i][iters > i]
        print('iter: {}, pri_resid: {:.5e}, dual_resid: {:.5e}, mu: {:.5e}'.format(
            i, row[:, 0].mean(), row[:, 1].mean(), row[:, 2].mean()))


def f64_arithmetic_serves(nz, nineq, neq, lib=None):
    """Sizes at which a float32 QP is solved in float64 ARITHMETIC (QPX_F32_WIDE, include/qpx.h: float32 tensors on the
    caller's side, float64 factors and arithmetic in the kernels, which widen on load and narrow on store): wherever
    the float64 matrix-core kernels serve the size -- the tile kernels (nineq <= 112, nz+neq+nineq <= 208) and, since
    round 4, the large-QP family.  On MI355X the f64 matrix-core loop is faster than the f32 thread-grid loop plus its
    finishing iterations, and its answer is the float64 solution of the float32 data.  The library is the authority
    (qpx_supported); without one the tile-kernel rule is returned."""
    if lib is not None:
        return lib.dll.qpx_kernel_family(_lib.QPX_F32_WIDE, nz, nineq, neq) in (_lib.FAMILY_TILE, _lib.FAMILY_BIG)
    return nineq <= 112 and nz + neq + nineq <= 208


def QPFunction(eps=1e-12, verbose=0, notImprovedLim=3,
               maxIter=20, solver=QPSolvers.PDIPM_BATCHED,
               check_Q_spd=True, refine=None):
    """`refine` is the one argument the reference does not have.  For float32 inputs:
      None (automatic) -- sizes the float64 tile kernels serve (f64_arithmetic_serves): the float64 kernels run on the
            float32 tensors (they widen on load and narrow results and gradients on store; the factors between
            forward and backward are float64); other sizes: as refine=2;
      0  -- the pure float32 kernels, nothing else (fastest at some sizes; the pre-computed products R = G Q^-1 G^T
            carry ~1e-2 relative error in float32 on the benchmark generator, so the loop kernel alone lands 20x
            further from the float64 answer than the reference's float32 run does);
      k > 0 -- the float32 kernels + k finishing Newton steps on the residuals of the ORIGINAL problem data
            (KKTFactors.polish -- the reference's KKTSolvers.IR_UNOPT idea, batch.py:244-270; each with one
            in-kernel refinement step per KKT solve, also applied to the backward solve).
    float64 inputs: None = 0.
    Memory: with refine=None a float32 batch in the large-QP family keeps a float64 factor blob (9.4 MB per QP at
    nz = nineq = 500, twice the float32 family's); a batch that only fits HBM with float32 factors should pass refine=2
    (float32 kernels + finishing iterations) or refine=0 explicitly."""
    class QPFunctionFn(Function):
        @staticmethod
        def forward(ctx, Q_, p_, G_, h_, A_, b_):
            nBatch = extract_nBatch(Q_, p_, G_, h_, A_, b_)
            nineq, nz = G_.size(-2), G_.size(-1)
            neq = A_.size(-2) if A_.nelement() > 0 else 0
            # float32 data, float64 arithmetic (see QPFunction.__doc__)
            ctx.wide = (solver == QPSolvers.PDIPM_BATCHED and refine is None and Q_.dtype == torch.float32
                        and f64_arithmetic_serves(nz, nineq, neq, _lib.backend_for(Q_)))
            Q, _ = expandParam(Q_, nBatch, 3)
            p, _ = expandParam(p_, nBatch, 2)
            G, _ = expandParam(G_, nBatch, 3)
            h, _ = expandParam(h_, nBatch, 2)
            A, _ = expandParam(A_, nBatch, 3)
            b, _ = expandParam(b_, nBatch, 2)

            assert(neq > 0 or nineq > 0)
            ctx.neq, ctx.nineq, ctx.nz = neq, nineq, nz

            if solver == QPSolvers.PDIPM_BATCHED:
                fac = KKTFactors.build(Q, G, A, nBatch, wide=ctx.wide)   # qp.py:93
                res = fac.ipm(p, h, b, eps, maxIter, notImprovedLim,
                              want_trace=(verbose == 1))             # qp.py:94-96
                ctx.refine = (2 if Q.dtype == torch.float32 and not ctx.wide else 0) if refine is None else int(refine)
                if ctx.refine > 0:
                    # (the solves inside a finishing step are NOT refined: the step's own residuals are exact, and refining
                    # the directions as well changes nothing in the answer -- C2 / C3 float32, two steps: the same error
                    # distribution to three digits -- for 40 % more time per step; profiles/archive/r04f)
                    res = fac.polish(p, h, b, res, steps=ctx.refine, refine=0)
                # one small read-back: the reference raises here too (qp.py:81-85, batch.py:379-386)
                fac.raise_on_failure(check_Q_spd)
                if verbose == 1:
                    _print_trace(res)
                if verbose >= 0:
                    if not bool((res.best_resid <= 1.).all().item()):
                        print(pdipm_b.INACC_ERR)                     # batch.py:141-142,205-206
                ctx.fac = fac
                zhats, ctx.nus, ctx.lams, ctx.slacks = res.zhat, res.nu, res.lam, res.slacks
            elif solver == QPSolvers.CVXPY:
                # forward by an external CPU solver, backward by the HIP kernels (qp.py:97-120,142-143)
                from .solvers import external
                zhats, ctx.nus, ctx.lams, ctx.slacks = external.forward_batch(Q, p, G, h, A, b, neq)
                ctx.fac = None
                ctx.refine = 0 if refine is None else int(refine)
            else:
                assert False

            ctx.save_for_backward(zhats, Q_, p_, G_, h_, A_, b_)
            return zhats

        @staticmethod
        def backward(ctx, dl_dzhat):
            zhats, Q, p, G, h, A, b = ctx.saved_tensors
            nBatch = extract_nBatch(Q, p, G, h, A, b)
            Q, Q_e = expandParam(Q, nBatch, 3)
            p, p_e = expandParam(p, nBatch, 2)
            G, G_e = expandParam(G, nBatch, 3)
            h, h_e = expandParam(h, nBatch, 2)
            A, A_e = expandParam(A, nBatch, 3)
            b, b_e = expandParam(b, nBatch, 2)
            neq = ctx.neq

            fac = ctx.fac
            if fac is None:                                          # qp.py:142-143
                fac = KKTFactors.build(Q, G, A, nBatch)
                fac.raise_on_failure(check_Q_spd)

            # d = clamp(lams)/clamp(slacks), factor_kkt, solve_kkt(dl_dzhat, 0, 0, 0) and the outer
            # products (qp.py:148-173) happen inside one kernel.  Only the gradients autograd asks for are
            # formed (ctx.needs_input_grad), and the `.mean(0)` of a parameter the batch shares
            # (qp.py:159-177) is taken inside KKTFactors.backward -- for Q, G, A as one contraction over
            # the batch instead of nBatch outer products.
            want = tuple(ctx.needs_input_grad[:6])
            grads = fac.backward(zhats, ctx.lams, ctx.slacks, ctx.nus, dl_dzhat, want=want,
                                 shared=(Q_e, p_e, G_e, h_e, A_e, b_e),
                                 refine=1 if (ctx.refine > 0 and fac.refine_ok) else 0)
            if neq == 0:
                grads = grads[:4] + (None, None)
            return grads
    return QPFunctionFn.apply
