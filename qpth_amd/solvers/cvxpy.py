"""External-solver forward (QPSolvers.CVXPY, qpth/qp.py:97-120 + qpth/solvers/cvxpy.py:5-31).

Out of scope as a backend (it is a per-QP CPU loop in the reference as well); kept so that
QPFunction(solver=QPSolvers.CVXPY) still differentiates through the HIP backward kernel with
externally produced (zhat, nu, lam, slacks).  cvxpy is imported lazily -- `import qpth_amd`
never needs it (unlike the reference, whose qpth/solvers/__init__.py:3 imports it eagerly).
"""
import numpy as np
import torch


def forward_single_np(Q, p, G, h, A, b):
    import cvxpy as cp
    nz, neq, nineq = p.shape[0], A.shape[0] if A is not None else 0, G.shape[0]
    z_ = cp.Variable(nz)
    obj = cp.Minimize(0.5 * cp.quad_form(z_, Q) + p.T @ z_)
    eqCon = A @ z_ == b if neq > 0 else None
    slacks = cp.Variable(nineq)
    ineqCon = G @ z_ + slacks == h
    slacksCon = slacks >= 0
    cons = [x for x in [eqCon, ineqCon, slacksCon] if x is not None]
    prob = cp.Problem(obj, cons)
    prob.solve()
    zhat = np.array(z_.value).ravel()
    nu = np.array(eqCon.dual_value).ravel() if eqCon is not None else None
    lam = np.array(ineqCon.dual_value).ravel()
    slacks = np.array(slacks.value).ravel()
    return prob.value, zhat, nu, lam, slacks


def forward_batch(Q, p, G, h, A, b, neq):
    nBatch = Q.size(0)
    outs = {k: [] for k in ("z", "nu", "lam", "s")}
    for i in range(nBatch):
        Ai, bi = (A[i], b[i]) if neq > 0 else (None, None)
        _, zi, nui, lami, si = forward_single_np(
            *[x.detach().cpu().numpy() if x is not None else None for x in (Q[i], p[i], G[i], h[i], Ai, bi)])
        outs["z"].append(zi); outs["lam"].append(lami); outs["s"].append(si)
        outs["nu"].append(nui if neq > 0 else np.zeros(0))
    mk = lambda v: torch.tensor(np.stack(v), dtype=Q.dtype, device=Q.device)  # noqa: E731
    return mk(outs["z"]), mk(outs["nu"]), mk(outs["lam"]), mk(outs["s"])
