"""ctypes front-end of oracle/qp_oracle.c -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's PDIPM path
(qpth/solvers/pdipm/batch.py:47-470, qpth/qp.py:127-182).  It exists to *check* the HIP
path: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
this module.  Nothing under qpth_amd/ does.

Parity pinning: tests/test_oracle_golden.py compares this oracle with outputs of the
reference itself (tests/golden/*.npz, produced by tests/golden/make_golden.py in the build
container, where /root/reference is importable).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqp_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement (gcc, two dtypes).  Building the checker is not using it."""
    src = os.path.join(_HERE, "qp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for sfx in ("_f64", "_f32"):
            getattr(_lib, "qpo_state_bytes" + sfx).restype = ctypes.c_size_t
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "_f64"
    if dtype == np.float32:
        return "_f32"
    raise TypeError("oracle supports float32/float64, got %s" % dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _bcast(X, B, nd, dtype):
    """qpth/util.py:44-50 expandParam, materialised (the C code wants dense batches)."""
    X = np.asarray(X, dtype=dtype)
    if X.size == 0:
        return None
    if X.ndim == nd:
        return np.ascontiguousarray(X)
    if X.ndim == nd - 1:
        return np.ascontiguousarray(np.broadcast_to(X[None], (B,) + X.shape))
    raise RuntimeError("Unexpected number of dimensions.")


class OracleQP:
    """One pre-factored batch (the reference's Q_LU, S_LU, R of batch.py:375-429)."""

    def __init__(self, Q, p, G, h, A=None, b=None, nthreads=None):
        dims = [3, 2, 3, 2, 3, 2]
        params = [Q, p, G, h, A if A is not None else np.zeros(0), b if b is not None else np.zeros(0)]
        B = 1
        for X, dmn in zip(params, dims):                  # qpth/util.py:53-59 extract_nBatch
            if np.ndim(X) == dmn:
                B = np.shape(X)[0]
                break
        dtype = np.asarray(Q).dtype
        self.dtype, self.sfx, self.B = dtype, _sfx(dtype), B
        self.Q = _bcast(Q, B, 3, dtype)
        self.p = _bcast(p, B, 2, dtype)
        self.G = _bcast(G, B, 3, dtype)
        self.h = _bcast(h, B, 2, dtype)
        self.A = _bcast(params[4], B, 3, dtype)
        self.b = _bcast(params[5], B, 2, dtype)
        self.n = self.Q.shape[1]
        self.m = self.G.shape[1]
        self.q = 0 if self.A is None else self.A.shape[1]
        self.nthreads = int(nthreads or os.cpu_count() or 1)
        L = lib()
        nbytes = getattr(L, "qpo_state_bytes" + self.sfx)(B, self.n, self.m, self.q)
        self.state = np.zeros(nbytes + 16, dtype=np.uint8)
        self.err = np.zeros(B, dtype=np.int32)
        nfail = getattr(L, "qpo_pre_factor" + self.sfx)(
            B, self.n, self.m, self.q, _p(self.Q), _p(self.G), _p(self.A), _p(self.state),
            _p(self.err), self.nthreads)
        if nfail:
            raise RuntimeError("qpth Error: Cannot perform LU factorization on Q.")  # batch.py:381-386

    # -- batch.py:435-470 -------------------------------------------------------------
    def factor_kkt(self, d):
        d = _c(d, self.dtype)
        return getattr(lib(), "qpo_factor_kkt" + self.sfx)(
            self.B, self.n, self.m, self.q, _p(self.state), _p(d), self.nthreads)

    # -- batch.py:349-372 -------------------------------------------------------------
    def solve_kkt(self, d, rx, rs, rz, ry=None):
        dt = self.dtype
        d, rx, rs, rz = (_c(v, dt) for v in (d, rx, rs, rz))
        ry = _c(ry, dt) if self.q > 0 else None
        dx = np.zeros((self.B, self.n), dt); ds = np.zeros((self.B, self.m), dt)
        dz = np.zeros((self.B, self.m), dt); dy = np.zeros((self.B, self.q), dt)
        getattr(lib(), "qpo_solve_kkt" + self.sfx)(
            self.B, self.n, self.m, self.q, _p(self.state), _p(self.G), _p(self.A), _p(d), _p(rx),
            _p(rs), _p(rz), _p(ry), _p(dx), _p(ds), _p(dz), _p(dy) if self.q > 0 else None,
            self.nthreads)
        return dx, ds, dz, (dy if self.q > 0 else None)

    # -- batch.py:47-207 ---------------------------------------------------------------
    def forward(self, eps=1e-12, maxIter=20, notImprovedLim=3, per_qp=False, stall_policy=2,
                want_trace=False):
        dt = self.dtype
        B, n, m, q = self.B, self.n, self.m, self.q
        x = np.zeros((B, n), dt); y = np.zeros((B, q), dt)
        z = np.zeros((B, m), dt); s = np.zeros((B, m), dt)
        iters = np.zeros(B, np.int32); best = np.zeros(B, dt)
        trace = np.full((maxIter, 3), np.nan, dt) if want_trace else None
        trips = getattr(lib(), "qpo_forward" + self.sfx)(
            B, n, m, q, _p(self.Q), _p(self.p), _p(self.G), _p(self.h), _p(self.A), _p(self.b),
            _p(self.state), ctypes.c_double(eps), int(maxIter), int(notImprovedLim), int(bool(per_qp)),
            int(stall_policy), _p(x), _p(y) if q > 0 else None, _p(z), _p(s), _p(iters), _p(best),
            _p(trace), self.nthreads)
        info = dict(iters=iters, best_resid=best, trips=trips, trace=trace)
        return x, (y if q > 0 else None), z, s, info

    # -- qp.py:127-182 -----------------------------------------------------------------
    def backward(self, zhat, lam, slack, nu, dl_dz):
        dt = self.dtype
        B, n, m, q = self.B, self.n, self.m, self.q
        zhat, lam, slack, dl_dz = (_c(v, dt) for v in (zhat, lam, slack, dl_dz))
        nu = _c(nu, dt) if q > 0 else None
        dQ = np.zeros((B, n, n), dt); dp = np.zeros((B, n), dt)
        dG = np.zeros((B, m, n), dt); dh = np.zeros((B, m), dt)
        dA = np.zeros((B, q, n), dt) if q > 0 else None
        db = np.zeros((B, q), dt) if q > 0 else None
        getattr(lib(), "qpo_backward" + self.sfx)(
            B, n, m, q, _p(self.G), _p(self.A), _p(self.state), _p(zhat), _p(lam), _p(slack), _p(nu),
            _p(dl_dz), _p(dQ), _p(dp), _p(dG), _p(dh), _p(dA), _p(db), self.nthreads)
        return dQ, dp, dG, dh, dA, db


def qp_forward_backward(Q, p, G, h, A, b, dl_dz=None, eps=1e-12, maxIter=20, notImprovedLim=3,
                        per_qp=False, stall_policy=2, nthreads=None):
    """QPFunction(...)(Q,p,G,h,A,b) then .backward(dl_dz) as the reference runs them
    (qp.py:92-96, 127-182); returns (zhat, nu, lam, slacks, grads-or-None, info).
    Gradients of broadcast (un-batched) parameters are mean-reduced as qp.py:159-177 does."""
    o = OracleQP(Q, p, G, h, A, b, nthreads=nthreads)
    x, y, z, s, info = o.forward(eps, maxIter, notImprovedLim, per_qp, stall_policy)
    grads = None
    if dl_dz is not None:
        dQ, dp, dG, dh, dA, db = o.backward(x, z, s, y, dl_dz)
        raw = [Q, p, G, h, A, b]
        nd = [3, 2, 3, 2, 3, 2]
        out = []
        for g, r, k in zip((dQ, dp, dG, dh, dA, db), raw, nd):
            if g is None:
                out.append(None)
            elif r is not None and np.ndim(r) == k - 1:
                out.append(g.mean(0))
            else:
                out.append(g)
        grads = tuple(out)
    return x, y, z, s, grads, info
