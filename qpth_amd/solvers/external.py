"""Forward by an EXTERNAL solver, backward by the HIP kernels.

The reference's `QPSolvers.CVXPY` (qpth/qp.py:97-120) solves every QP of the batch on the CPU with
cvxpy and then differentiates with the same factor_kkt + solve_kkt backward as the PDIPM path
(qp.py:142-155, `ctx.Q_LU` rebuilt because the external solver left no factors).  Here that is:
any callable `solve(Q, p, G, h, A, b) -> (zhat, nu, lam, slacks)` on numpy arrays of ONE QP
(`A`, `b`, `nu` are None without equality constraints), registered with `set_solver`; its outputs are
moved to the device and `qpx_backward` does the rest.  With nothing registered a small cvxpy model is
used if cvxpy is importable (it is not in the build image: tests register a stub that replays
reference-produced solutions, tests/test_gpu_parity.py::test_backward_from_external_solutions).
"""
import numpy as np
import torch

_SOLVER = None


def set_solver(fn):
    """Register `fn(Q, p, G, h, A, b) -> (zhat, nu, lam, slacks)` (numpy, one QP); None restores cvxpy."""
    global _SOLVER
    _SOLVER = fn


def _cvxpy_solve(Q, p, G, h, A, b):
    try:
        import cvxpy as cp
    except ImportError as e:                                    # loud: there is no silent fallback
        raise RuntimeError("QPSolvers.CVXPY needs cvxpy (not installed) or a solver registered with "
                           "qpth_amd.solvers.external.set_solver()") from e
    z = cp.Variable(p.shape[0])
    ineq = G @ z <= h
    cons = [ineq]
    eq = None
    if A is not None:
        eq = A @ z == b
        cons.append(eq)
    prob = cp.Problem(cp.Minimize(0.5 * cp.quad_form(z, cp.psd_wrap(Q)) + p @ z), cons)
    prob.solve()
    if prob.status not in ("optimal", "optimal_inaccurate"):
        raise RuntimeError("external solver: QP is %s" % prob.status)
    zhat = np.asarray(z.value).ravel()
    return zhat, (np.asarray(eq.dual_value).ravel() if eq is not None else None), \
        np.asarray(ineq.dual_value).ravel(), h - G @ zhat


def forward_batch(Q, p, G, h, A, b, neq):
    """(B, ...) device tensors -> zhat (B,n), nu (B,neq), lam (B,m), slacks (B,m) on the same device."""
    solve = _SOLVER or _cvxpy_solve
    host = [x.detach().cpu().numpy() for x in (Q, p, G, h)]
    hA, hb = (A.detach().cpu().numpy(), b.detach().cpu().numpy()) if neq > 0 else (None, None)
    cols = [[], [], [], []]
    for i in range(host[0].shape[0]):
        out = solve(host[0][i], host[1][i], host[2][i], host[3][i],
                    hA[i] if neq > 0 else None, hb[i] if neq > 0 else None)
        zi, nui, lami, si = out
        for c, v in zip(cols, (zi, nui if neq > 0 else np.zeros(0), lami, si)):
            c.append(np.asarray(v, dtype=host[0].dtype).ravel())
    return tuple(torch.as_tensor(np.stack(c), dtype=Q.dtype, device=Q.device) for c in cols)
