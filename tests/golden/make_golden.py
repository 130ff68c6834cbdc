#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (locuslab/qpth at /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

The reference is pure Python on PyTorch; plain `import qpth` fails here because
qpth/solvers/__init__.py:3 imports cvxpy (not installed), so an empty in-memory `cvxpy`
module is registered first -- nothing of cvxpy is used by the PDIPM path.  The reference is
imported unmodified and run on the PyTorch CPU backend.

Each fixture stores the outputs of the reference for seeded inputs from tests/problems.py
(small inputs are stored too; large ones are regenerated from the seed and verified through a
stored checksum):

  zhat, nu, lam, slacks ... qpth.qp.QPFunction forward (qp.py:92-96) on the whole batch
  b1_*  .................... the same, each QP solved alone (batch of one)
  dQ, dp, dG, dh, dA, db ... gradients of QPFunction.backward (qp.py:127-182)
  kkt_* .................... pre_factor_kkt + factor_kkt + solve_kkt, and factor_solve_kkt
                              (test.py:222-234 test_lu_kkt_solver)
"""
import os
import sys
import types

import numpy as np

sys.modules.setdefault("cvxpy", types.ModuleType("cvxpy"))
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
import qpth  # noqa: E402
from qpth.qp import QPFunction  # noqa: E402
import qpth.solvers.pdipm.batch as pdipm_b  # noqa: E402
from qpth.util import bdiag, expandParam, extract_nBatch  # noqa: E402

import problems  # noqa: E402

torch.set_num_threads(os.cpu_count() or 1)


def t(x, dtype):
    x = np.asarray(x)
    return torch.tensor(x.astype(dtype)) if x.size else torch.empty(0, dtype=torch.from_numpy(np.zeros(1, dtype)).dtype)


def run_ref(Q, p, G, h, A, b, dl=None, dtype=np.float64, **kw):
    """QPFunction forward (+ backward) on the reference; returns dict of numpy outputs."""
    tq = [t(x, dtype) for x in (Q, p, G, h, A, b)]
    for x in tq:
        if x.nelement() > 0:
            x.requires_grad_(True)
    out = {}
    # qp.py stores nus/lams/slacks on ctx only; fetch them by driving the solver entry points
    # exactly as qp.py:92-96 does.
    with torch.no_grad():
        nB = extract_nBatch(*tq)
        e = [expandParam(x, nB, d)[0] for x, d in zip(tq, (3, 2, 3, 2, 3, 2))]
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(e[0], e[2], e[4])
        zh, nu, lam, sl = pdipm_b.forward(e[0], e[1], e[2], e[3], e[4], e[5], Q_LU, S_LU, R,
                                          kw.get("eps", 1e-12), -1, kw.get("notImprovedLim", 3),
                                          kw.get("maxIter", 20))
    out["zhat"] = zh.numpy().copy()
    out["lam"] = lam.numpy().copy()
    out["slacks"] = sl.numpy().copy()
    out["nu"] = nu.numpy().copy() if nu is not None else np.zeros((nB, 0), dtype)
    zhat = QPFunction(verbose=-1, **kw)(*tq)
    assert np.array_equal(zhat.detach().numpy(), out["zhat"])
    if dl is not None:
        dlt = t(dl(out["zhat"]) if callable(dl) else dl, dtype)
        zhat.backward(dlt)
        out["dl_dz"] = dlt.numpy().copy()
        for name, x in zip(("dQ", "dp", "dG", "dh", "dA", "db"), tq):
            if x.nelement() > 0:
                out[name] = x.grad.numpy().copy()
    return out


def run_ref_b1(Q, p, G, h, A, b, dtype=np.float64):
    """Each QP alone (batch of one): what a per-QP kernel must agree with."""
    B = Q.shape[0]
    acc = {k: [] for k in ("zhat", "lam", "slacks", "nu")}
    for i in range(B):
        sl = lambda x: x[i:i + 1] if np.asarray(x).size else x  # noqa: E731
        o = run_ref(sl(Q), sl(p), sl(G), sl(h), sl(A), sl(b), dtype=dtype)
        for k in acc:
            acc[k].append(o[k][0])
    return {"b1_" + k: np.stack(v) for k, v in acc.items()}


def checksum(*arrs):
    return np.array([float(np.sum(np.asarray(a, np.float64) * np.cos(np.arange(np.asarray(a).size).reshape(np.shape(a)) % 97)))
                     for a in arrs if np.asarray(a).size])


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def main():
    # 1. the five gradient-test problems of test.py:99-187 (B=1, nz=10, f64)
    for name, (neq, nineq, qs, gs, as_) in problems.GRADS_CASES.items():
        Q, p, G, h, A, b, truez = problems.grads_qp(10, neq, nineq, qs, gs, as_)
        if neq == 0:
            A = np.zeros(0); b = np.zeros(0)
        o = run_ref(Q, p, G, h, A, b, dl=lambda z: z - truez)   # test.py:88
        save("grads_" + name, Q=Q, p=p, G=G, h=h, A=A, b=b, truez=truez, **o)

    # 2. test.py:190-234 KKT-solver cross-check (B=2, nx=5, nineq=4, neq=3)
    Q, p, G, h, A, b, d, rx, rs, rz, ry = problems.kkt_problem(seed=0)
    tq = [torch.tensor(x) for x in (Q, p, G, h, A, b)]
    nB = extract_nBatch(*tq)
    e = [expandParam(x, nB, k)[0] for x, k in zip(tq, (3, 2, 3, 2, 3, 2))]
    td, trx, trs, trz, try_ = [torch.tensor(x) for x in (d, rx, rs, rz, ry)]
    dx, ds, dz, dy = pdipm_b.factor_solve_kkt(e[0], bdiag(td), e[2], e[4], trx, trs, trz, try_)
    Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(e[0], e[2], e[4])
    pdipm_b.factor_kkt(S_LU, R, td)
    dx_, ds_, dz_, dy_ = pdipm_b.solve_kkt(Q_LU, td, e[2], e[4], S_LU, trx, trs, trz, try_)
    save("kkt_solver", Q=Q, p=p, G=G, h=h, A=A, b=b, d=d, rx=rx, rs=rs, rz=rz, ry=ry,
         full_dx=dx.numpy(), full_ds=ds.numpy(), full_dz=dz.numpy(), full_dy=dy.numpy(),
         dx=dx_.numpy(), ds=ds_.numpy(), dz=dz_.numpy(), dy=dy_.numpy(), R=R.numpy())

    # 3. BASELINE.json configs, prof-linear.py generator
    def prof_case(name, B, n, m, q, seed, dtype=np.float64, store_inputs=True, b1=True):
        Q, p, G, h, A, b = problems.prof_qp(B, n, m, q, seed, dtype)
        o = run_ref(Q, p, G, h, A, b, dl=lambda z: np.ones_like(z), dtype=dtype)
        if b1:
            o.update(run_ref_b1(Q, p, G, h, A, b, dtype=dtype))
        extra = dict(Q=Q, p=p, G=G, h=h, A=A, b=b) if store_inputs else {}
        save(name, shape=np.array([B, n, m, q, seed]), input_checksum=checksum(Q, p, G, h, A, b),
             **extra, **o)

    prof_case("c1_b8_n10_m5_f64", 8, 10, 5, 0, 0)
    prof_case("c1_b8_n10_m5_f32", 8, 10, 5, 0, 0, np.float32)
    prof_case("c3s_b4_n20_m10_q4_f64", 4, 20, 10, 4, 1)
    prof_case("c2s_b4_n100_m100_f64", 4, 100, 100, 0, 0, store_inputs=False)
    prof_case("c3s_b4_n100_m50_q10_f64", 4, 100, 50, 10, 0, store_inputs=False)
    prof_case("c5s_b6_n64_m64_f64", 6, 64, 64, 0, 2, store_inputs=False)

    # 4. broadcast (un-batched) parameters: Q, G, A shared; p, h, b batched (util.py:44-59,
    #    mean-reduced grads qp.py:159-177)
    Q, p, G, h, A, b = problems.random_dense_qp(5, 12, 9, 3, seed=7)
    r = np.random.RandomState(3)
    Qs, Gs, As = Q[0], G[0], A[0]
    z0 = r.randn(5, 12)
    hs = np.einsum("mn,bn->bm", Gs, z0) + r.rand(5, 9) + 0.1
    bs = np.einsum("qn,bn->bq", As, z0)
    dl = r.randn(5, 12)
    o = run_ref(Qs, p, Gs, hs, As, bs, dl=dl)
    save("broadcast_b5_n12_m9_q3", Q=Qs, p=p, G=Gs, h=hs, A=As, b=bs, **o)

    # 5. everything un-batched (nBatch = 1 path of util.py:53-59)
    o = run_ref(Qs, p[0], Gs, hs[0], As, bs[0], dl=dl[:1])
    save("unbatched_n12_m9_q3", Q=Qs, p=p[0], G=Gs, h=hs[0], A=As, b=bs[0], **o)


def extra():
    """Edge shapes added after the first fixture set (run with --extra; leaves the other files alone):
    one variable / one constraint, neq = nz - 1 (one degree of freedom left), duplicated inequality rows
    (R + D^-1 is still SPD, R itself is rank deficient)."""
    def case(name, arrs):
        Q, p, G, h, A, b = arrs
        o = run_ref(Q, p, G, h, A, b, dl=lambda z: np.ones_like(z))
        o.update(run_ref_b1(Q, p, G, h, A, b))
        save(name, Q=Q, p=p, G=G, h=h, A=A, b=b, **o)

    case("edge_b3_n1_m1_q0", problems.prof_qp(3, 1, 1, 0, 5))
    case("edge_b2_n6_m4_q5", problems.prof_qp(2, 6, 4, 5, 6))
    Q, p, G, h, A, b = problems.prof_qp(2, 8, 5, 0, 7)
    case("edge_dup_b2_n8_m10_q0", (Q, p, np.concatenate([G, G], 1), np.concatenate([h, h], 1), A, b))


def callers():
    """Round 2 (run with --callers): the shapes the reference's own callers use, with the
    parameters SHARED by the batch exactly as the notebooks pass them, and f32/f64 pairs of the prof-linear
    workload from which the f32 tests take the reference's own f32-vs-f64 error distribution.

      cls_*     example-cls-layer.ipynb cell 3: Q = L L^T + eps I (nCls x nCls, shared), p = the layer input
                (batched), G (nineq x nCls, shared), h = G z0 + s0 (shared), no equalities; f32 there
      sudoku_*  example-sudoku.ipynb cell 10: Q = 0.1 I, G = -I, h = 0, A (40 x 64, shared parameter),
                b = 1 (all shared), p = -puzzle (batched), f64
      f32pair_* prof-linear.py generator: zhat of the reference in f32 and in f64 on the same QPs
    """
    # torch.linalg.lu_factor with more than one thread hangs in this container's MKL for matrices of order
    # >= 160 (observed: "oneMKL ERROR: Parameter 6 was incorrect on entry to DLASWP", then a dead-lock)
    torch.set_num_threads(1)
    r = np.random.RandomState(11)
    nCls, nineq, B = 2, 200, 32
    L = np.tril(r.rand(nCls, nCls))
    Q = L.dot(L.T) + 1e-4 * np.eye(nCls)
    G = r.uniform(-1, 1, (nineq, nCls))
    h = G.dot(np.zeros(nCls)) + np.ones(nineq)
    x = np.maximum(r.randn(B, nCls), 0) * 3.0                       # the relu'd features the layer feeds in as p
    e = np.zeros(0)
    dl = r.randn(B, nCls)
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        o = run_ref(Q, x, G, h, e, e, dl=dl, dtype=dt)
        save("cls_b32_n2_m200_" + tag, Q=Q, p=x, G=G, h=h, A=e, b=e, **o)

    nx, neq, B = 64, 40, 16
    Q = 0.1 * np.eye(nx)
    G = -np.eye(nx)
    h = np.zeros(nx)
    A = r.rand(neq, nx)
    b = np.ones(neq)
    puzzles = (r.rand(B, nx) < 0.25).astype(np.float64)
    dl = r.randn(B, nx)
    o = run_ref(Q, -puzzles, G, h, A, b, dl=dl)
    save("sudoku_b16_n64_m64_q40_f64", Q=Q, p=-puzzles, G=G, h=h, A=A, b=b, **o)

    def pair(name, B, n, m, q, seed):
        arrs64 = problems.prof_qp(B, n, m, q, seed, np.float64)
        arrs32 = problems.prof_qp(B, n, m, q, seed, np.float32)
        o64 = run_ref(*arrs64, dtype=np.float64)
        o32 = run_ref(*arrs32, dtype=np.float32)
        save(name, shape=np.array([B, n, m, q, seed]), input_checksum=checksum(*arrs64),
             zhat_f64=o64["zhat"], zhat_f32=o32["zhat"])

    pair("f32pair_c2_b32_n100_m100", 32, 100, 100, 0, 3)
    pair("f32pair_c3_b32_n100_m50_q10", 32, 100, 50, 10, 4)


if __name__ == "__main__":
    if "--extra" in sys.argv:
        extra()
    elif "--callers" in sys.argv:
        callers()
    else:
        main()
