"""ctypes binding of the C ABI in include/qpx.h (libqpx_hip.so).

The product path is MI355X only: `hip()` loads qpth_amd/libqpx_hip.so (built by
`__graft_entry__.build()` / qpth_amd/csrc/build.sh) and FAILS LOUDLY when it is missing or
when tensors are not on a HIP device -- there is no CPU fallback anywhere in this package.

`QpxLib(path)` is the marshalling layer: torch tensors in, raw pointers + batch strides +
the current HIP stream out.  Tests additionally instantiate it over tests/emu's host-thread
emulation of the very same kernel bodies (`set_test_backend`), which exists so that the
GPU-less CI can exercise the host logic; nothing in the package selects it by itself.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(_HERE, "libqpx_hip.so")

QPX_F32, QPX_F64, QPX_F32_WIDE = 0, 1, 2      # QPX_F32_WIDE: float32 arrays, float64 factors and arithmetic (include/qpx.h)
ST_Q_NOT_SPD, ST_A_RANK, ST_KKT_BREAKDOWN, ST_INACCURATE, ST_MAXITER, ST_NONFINITE = 1, 2, 4, 8, 16, 32
STALL_OFF, STALL_REFERENCE, STALL_FLOOR = 0, 1, 2
FAMILY_GRID, FAMILY_TILE, FAMILY_BIG = 1, 2, 3

_vp, _i, _i64, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double

_SIGNATURES = {
    # name: (restype, argtypes)       -- must list every symbol include/qpx.h declares
    "qpx_abi_version": (_i, []),
    "qpx_strerror": (ctypes.c_char_p, [_i]),
    "qpx_factor_elems": (ctypes.c_size_t, [_i, _i, _i, _i]),
    "qpx_max_dim": (_i, []),
    "qpx_supported": (_i, [_i, _i, _i, _i]),
    "qpx_kernel_family": (_i, [_i, _i, _i, _i]),
    "qpx_refine_supported": (_i, [_i, _i, _i, _i]),
    "qpx_set_ipm_variant": (_i, [_i]),
    "qpx_get_ipm_variant": (_i, []),
    "qpx_can_share_factors": (_i, [_i, _i, _i, _i]),
    "qpx_big_gemm_r": (_i, [_i, _i, _i, _i, _i, _vp, _vp]),
    "qpx_pre_factor": (_i, [_i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "qpx_ipm": (_i, [_i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _d, _i, _i, _i,
                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qpx_forward": (_i, [_i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64,
                         _vp, _i64, _vp, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qpx_factor_solve_kkt": (_i, [_i, _i, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "qpx_backward": (_i, [_i, _i, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp, _vp, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "qpx_polish_supported": (_i, [_i, _i, _i, _i]),
    "qpx_polish": (_i, [_i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64,
                        _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "qpx_batch_outer": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _d, _vp, _vp, ctypes.c_size_t, _vp]),
    "qpx_batch_outer_workspace_elems": (ctypes.c_size_t, [_i, _i, _i, _i]),
    "qpx_dense_solve": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
}
ABI_SYMBOLS = tuple(_SIGNATURES)


def _dtype_code(t):
    if t.dtype == torch.float64:
        return QPX_F64
    if t.dtype == torch.float32:
        return QPX_F32
    raise TypeError("qpth_amd supports float32 and float64 tensors, got %s" % t.dtype)


def _code(factors, wide):
    """dtype code of a call: QPX_F32_WIDE when the caller's arrays are float32 and `factors` is the float64 blob"""
    if wide:
        assert factors.dtype == torch.float64
        return QPX_F32_WIDE
    return _dtype_code(factors)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


class Param:
    """A QP parameter as the kernels see it: dense trailing dims + a batch stride (0 = shared)."""

    def __init__(self, X, nbatch_dims):
        if X is None or X.nelement() == 0:
            self.t, self.stride = None, 0
            return
        if X.dim() == nbatch_dims:       # batched; an expand()ed view has stride 0 already
            if X.stride(0) == 0:
                X = X[0]
            else:
                X = X.contiguous()
                self.t, self.stride = X, (X.stride(0) if X.size(0) > 1 else 0)
                return
        self.t, self.stride = X.contiguous(), 0

    @property
    def ptr(self):
        return _ptr(self.t)


class QpxLib:
    def __init__(self, path, strict=True):
        """strict=False (scripts/ab_bench.py only): an older build of the library -- symbols it lacks are skipped and
        its ABI version is not checked; the product path always loads strictly."""
        if not os.path.exists(path):
            raise RuntimeError(
                "qpth_amd: %s is missing. Build the gfx950 extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or qpth_amd/csrc/build.sh). "
                "There is no CPU fallback." % path)
        self.path = path
        self.dll = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            if not strict and not hasattr(self.dll, name):
                continue
            fn = getattr(self.dll, name)       # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        if strict and self.dll.qpx_abi_version() != 8:
            raise RuntimeError("qpth_amd: ABI version mismatch in %s" % path)

    def check(self, code):
        if code != 0:
            raise RuntimeError("qpth_amd: %s (code %d)" % (self.dll.qpx_strerror(code).decode(), code))

    def factor_elems(self, dtype_code, n, m, q):
        return int(self.dll.qpx_factor_elems(dtype_code, n, m, q))

    # -- batch.py:375-429 ---------------------------------------------------------------
    def pre_factor(self, B, n, m, q, Q, G, A, factors, status, wide=False):
        Qp, Gp, Ap = Param(Q, 3), Param(G, 3), Param(A, 3)
        self.check(self.dll.qpx_pre_factor(
            _code(factors, wide), B, n, m, q, Qp.ptr, Qp.stride, Gp.ptr, Gp.stride, Ap.ptr, Ap.stride,
            _ptr(factors), _ptr(status), _stream(factors)))

    # -- batch.py:47-207 ----------------------------------------------------------------
    def ipm(self, B, n, m, q, p, h, b, factors, sfac, eps, maxIter, notImprovedLim, stall_policy,
            zhat, nu, lam, slack, iters, status, best_resid, trace=None, wide=False):
        pp, hp, bp = Param(p, 2), Param(h, 2), Param(b, 2)
        self.check(self.dll.qpx_ipm(
            _code(factors, wide), B, n, m, q, pp.ptr, pp.stride, hp.ptr, hp.stride, bp.ptr, bp.stride,
            _ptr(factors), int(sfac), float(eps), int(maxIter), int(notImprovedLim), int(stall_policy),
            _ptr(zhat), _ptr(nu), _ptr(lam), _ptr(slack), _ptr(iters), _ptr(status), _ptr(best_resid),
            _ptr(trace), _stream(factors)))

    # -- qp.py:92-96 ---------------------------------------------------------------------
    def forward(self, B, n, m, q, Q, p, G, h, A, b, factors, eps, maxIter, notImprovedLim,
                stall_policy, zhat, nu, lam, slack, iters, status, best_resid, trace=None, wide=False):
        Qp, Gp, Ap = Param(Q, 3), Param(G, 3), Param(A, 3)
        pp, hp, bp = Param(p, 2), Param(h, 2), Param(b, 2)
        self.check(self.dll.qpx_forward(
            _code(factors, wide), B, n, m, q, Qp.ptr, Qp.stride, pp.ptr, pp.stride, Gp.ptr, Gp.stride,
            hp.ptr, hp.stride, Ap.ptr, Ap.stride, bp.ptr, bp.stride, _ptr(factors), float(eps),
            int(maxIter), int(notImprovedLim), int(stall_policy), _ptr(zhat), _ptr(nu), _ptr(lam),
            _ptr(slack), _ptr(iters), _ptr(status), _ptr(best_resid), _ptr(trace), _stream(factors)))

    # -- batch.py:435-470 + 349-372 ------------------------------------------------------
    def factor_solve_kkt(self, B, n, m, q, factors, sfac, d, rx, rs, rz, ry, dx, ds, dz, dy, status,
                         refine=0, Q=None, G=None, A=None, wide=False):
        Qp, Gp, Ap = Param(Q, 3), Param(G, 3), Param(A, 3)
        self.check(self.dll.qpx_factor_solve_kkt(
            _code(factors, wide), B, n, m, q, _ptr(factors), int(sfac), _ptr(d), _ptr(rx), _ptr(rs), _ptr(rz),
            _ptr(ry), _ptr(dx), _ptr(ds), _ptr(dz), _ptr(dy), int(refine), Qp.ptr, Qp.stride, Gp.ptr, Gp.stride,
            Ap.ptr, Ap.stride, _ptr(status), _stream(factors)))

    # -- qp.py:127-182 --------------------------------------------------------------------
    def backward(self, B, n, m, q, factors, sfac, zhat, lam, slack, nu, dl_dz, dQ, dp, dG, dh, dA, db, status,
                 dx=None, dz=None, dy=None, refine=0, Q=None, G=None, A=None, wide=False):
        """Any of dQ..db may be None (gradient not wanted); dx, dz, dy: optional KKT solution outputs."""
        Qp, Gp, Ap = Param(Q, 3), Param(G, 3), Param(A, 3)
        self.check(self.dll.qpx_backward(
            _code(factors, wide), B, n, m, q, _ptr(factors), int(sfac), _ptr(zhat), _ptr(lam), _ptr(slack),
            _ptr(nu), _ptr(dl_dz), _ptr(dQ), _ptr(dp), _ptr(dG), _ptr(dh), _ptr(dA), _ptr(db),
            _ptr(dx), _ptr(dz), _ptr(dy), int(refine), Qp.ptr, Qp.stride, Gp.ptr, Gp.stride, Ap.ptr, Ap.stride,
            _ptr(status), _stream(factors)))

    # -- batch.py:92-198 in the original variables, as a finishing stage (KKTSolvers.IR_UNOPT, float32 refine=k)
    def polish(self, B, n, m, q, Q, p, G, h, A, b, factors, sfac, steps, refine, zhat, nu, lam, slack, best_resid, status):
        Qp, Gp, Ap = Param(Q, 3), Param(G, 3), Param(A, 3)
        pp, hp, bp = Param(p, 2), Param(h, 2), Param(b, 2)
        self.check(self.dll.qpx_polish(
            _dtype_code(factors), B, n, m, q, Qp.ptr, Qp.stride, pp.ptr, pp.stride, Gp.ptr, Gp.stride, hp.ptr, hp.stride,
            Ap.ptr, Ap.stride, bp.ptr, bp.stride, _ptr(factors), int(sfac), int(steps), int(refine),
            _ptr(zhat), _ptr(nu), _ptr(lam), _ptr(slack), _ptr(best_resid), _ptr(status), _stream(factors)))

    # -- qp.py:159-177, the `.mean(0)` of a shared parameter's gradient as one contraction over the batch
    def batch_outer(self, u, v, w, x, scale, out):
        """v is None: the batch mean of u's columns times `scale` (out: (r,)); w, x None: one product only"""
        B, r = u.shape
        c = v.shape[1] if v is not None else 1
        code = _dtype_code(out)
        # long batches: partial tiles per batch chunk in a workspace, summed in chunk order by a second launch
        need = int(self.dll.qpx_batch_outer_workspace_elems(code, B, r, c))
        ws = torch.empty(need, dtype=out.dtype, device=out.device) if need else None
        self.check(self.dll.qpx_batch_outer(code, B, r, c, _ptr(u), _ptr(v), _ptr(w), _ptr(x),
                                            float(scale), _ptr(out), _ptr(ws), need, _stream(out)))


    # -- the neq x neq correction of factor_solve_kkt_reg (batch.py:273-310): x = M^-1 r, M destroyed, r overwritten
    def dense_solve(self, M, rhs, status=None):
        B, k = rhs.shape
        assert M.shape == (B, k, k) and M.is_contiguous() and rhs.is_contiguous() and M.dtype == rhs.dtype
        self.check(self.dll.qpx_dense_solve(_dtype_code(rhs), B, k, _ptr(M), _ptr(rhs), _ptr(status), _stream(rhs)))
        return rhs


_HIP = None
_TEST_BACKEND = None


def hip():
    """The product backend: libqpx_hip.so.  Raises if it has not been built."""
    global _HIP
    if _HIP is None:
        _HIP = QpxLib(HIP_SO)
    return _HIP


def set_test_backend(lib):
    """tests/emu only: route calls to a QpxLib over the host-thread emulator (or None to undo)."""
    global _TEST_BACKEND
    _TEST_BACKEND = lib


def backend_for(t):
    """Backend that must execute work on tensor `t`.  CPU tensors are refused unless a test has
    explicitly installed the emulator."""
    if _TEST_BACKEND is not None:
        return _TEST_BACKEND
    if not t.is_cuda:
        raise RuntimeError(
            "qpth_amd runs on AMD Instinct GPUs only (got a %s tensor); there is no CPU fallback. "
            "Move Q, p, G, h, A, b to a HIP device." % t.device)
    return hip()
