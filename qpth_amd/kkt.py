"""Device-resident KKT factor state of one batch and the launches that use it.

`KKTFactors` is what the reference keeps as (Q_LU, S_LU, R) on ctx between forward and
backward (qpth/qp.py:93,150-155): here one HBM blob per QP written by qpx_pre_factor
(layout: qpth_amd/csrc/qpx_layout.h) plus the recorded `d` of factor_kkt.  When Q, G and A
are all shared by the batch (un-batched parameters, qpth/util.py:44-50) the blob is built
once and every workgroup reads the same copy.
"""
import contextlib
import threading

import numpy as np
import torch

from . import _lib

_STALL_POLICY = None     # None = automatic (reference counter for B == 1, floor rule otherwise)


def set_stall_policy(policy):
    """Override how `notImprovedLim` is applied per QP (see include/qpx.h); None = automatic."""
    global _STALL_POLICY
    assert policy in (None, _lib.STALL_OFF, _lib.STALL_REFERENCE, _lib.STALL_FLOOR)
    _STALL_POLICY = policy


def default_stall_policy(B):
    if _STALL_POLICY is not None:
        return _STALL_POLICY
    return _lib.STALL_REFERENCE if B == 1 else _lib.STALL_FLOOR


class IpmResult:
    __slots__ = ("zhat", "nu", "lam", "slacks", "iters", "status", "best_resid", "trace")


def _batch_of(*params3):
    for X in params3:
        if X is not None and X.nelement() > 0 and X.dim() == 3:
            return X.size(0)
    return 1


def _is_shared(X, B):
    if X is None or X.nelement() == 0:
        return True
    return X.dim() == 2 or X.stride(0) == 0 or (B > 1 and X.size(0) == 1)


class _PinnedPool:
    """Pinned int32 host buffers for the one small D2H copy behind a pre-factorisation (the per-QP status words).  A buffer
    belongs to exactly ONE KKTFactors from `take` until that object has read it (raise_on_failure) or dies; only then can
    another build() get it -- nothing is handed out twice however many builds are outstanding, and two host threads may
    build at once (autograd runs backward on threads of its own).  Allocating pinned memory per call would cost more than
    the kernels the asynchronous copy lets the host overlap, hence the free lists."""

    def __init__(self):
        self._lock = threading.Lock()
        self._free = {}
        self._pending = []          # (event, key, buffer): given up by their owner while the copy was still in flight

    @staticmethod
    def _done(ev):
        """True / False: the event has / has not completed; None: it cannot be asked (recorded inside a stream capture):
        the buffer behind it is dropped rather than handed out"""
        try:
            return bool(ev.query())
        except Exception:
            return None

    def take(self, device, count, may_allocate=True):
        key = (device.type, device.index, int(count))
        with self._lock:
            if self._pending:
                still = []
                for ev, k, buf in self._pending:
                    done = self._done(ev)
                    if done:
                        self._free.setdefault(k, []).append(buf)
                    elif done is not None:
                        still.append((ev, k, buf))
                self._pending = still
            lst = self._free.get(key)
            if lst:
                return key, lst.pop()
        if not may_allocate:
            return key, None
        return key, torch.zeros(count, dtype=torch.int32).pin_memory()

    def give(self, key, buf, event=None, keep=32):
        """event: the copy into `buf` may still be in flight -- the buffer is free once the event has completed"""
        with self._lock:
            done = True if event is None else self._done(event)
            if not done:
                if done is not None and len(self._pending) < keep:
                    self._pending.append((event, key, buf))
                return
            lst = self._free.setdefault(key, [])
            if len(lst) < keep:
                lst.append(buf)


_PINNED = _PinnedPool()


class KKTFactors:
    @classmethod
    def build(cls, Q, G, A, nBatch=None, wide=False):
        """pre_factor_kkt(Q, G, A)   (batch.py:375-429); enqueues one kernel, no host sync.
        wide: float32 tensors, float64 factors and arithmetic (QPX_F32_WIDE, include/qpx.h): every later call on these
        factors takes and returns float32 tensors, the blob is float64."""
        self = cls()
        self.wide = bool(wide)
        if self.wide and Q.dtype != torch.float32:
            raise TypeError("qpth_amd: wide=True is for float32 tensors")
        B = nBatch if nBatch is not None else _batch_of(Q, G, A)
        self.B = B
        self.n = Q.size(-1)
        self.m = G.size(-2)
        self.q = A.size(-2) if (A is not None and A.nelement() > 0) else 0
        if G.size(-1) != self.n or Q.size(-2) != self.n or (self.q and A.size(-1) != self.n):
            raise RuntimeError("qpth_amd: inconsistent QP sizes Q%s G%s A%s" % (
                tuple(Q.shape), tuple(G.shape), tuple(A.shape) if A is not None else ()))
        for X, what in ((Q, "Q"), (G, "G"), (A if self.q else None, "A")):
            if X is None:
                continue
            if X.dtype != Q.dtype or X.device != Q.device:
                raise RuntimeError("qpth_amd: %s is %s on %s but Q is %s on %s (all of Q, p, G, h, A, b must share one "
                                   "dtype and one device)" % (what, X.dtype, X.device, Q.dtype, Q.device))
            if X.dim() not in (2, 3) or (X.dim() == 3 and X.size(0) not in (1, B)):
                raise RuntimeError("qpth_amd: %s has shape %s for a batch of %d" % (what, tuple(X.shape), B))
        self.lib = _lib.backend_for(Q)
        self.dtype, self.device = Q.dtype, Q.device
        self.Q, self.G, self.A = Q, G, (A if self.q else None)     # the original data: iterative refinement evaluates residuals with it
        code = _lib.QPX_F32_WIDE if self.wide else (_lib.QPX_F64 if Q.dtype == torch.float64 else _lib.QPX_F32)
        self.elems = self.lib.factor_elems(code, self.n, self.m, self.q)
        # the A/B knob of the library is per host thread and selects the blob layout: remember the value the
        # factors are built under and re-apply it around every later call on them (autograd runs backward
        # on its own thread)
        self.variant = int(self.lib.dll.qpx_get_ipm_variant())
        # does the kernel family that serves this size refine KKT solves in the kernel (qpx_factor_solve_kkt(..., refine))?
        # (a pre-v5 build loaded non-strictly by scripts/ab_bench.py has no such symbol: it ignored `refine` where it could not refine)
        self.refine_ok = bool(self.lib.dll.qpx_refine_supported(code, self.n, self.m, self.q)) if hasattr(self.lib.dll, "qpx_refine_supported") else True
        # ... and does it have the finishing stage as a kernel (qpx_polish)?  Asked here, under the knob the factors are built
        # with: the answer depends on the calling thread's knob, and polish() may be reached from another thread
        pcode = _lib.QPX_F64 if Q.dtype == torch.float64 else _lib.QPX_F32
        self.polish_ok = (not self.wide and hasattr(self.lib.dll, "qpx_polish_supported")
                          and bool(self.lib.dll.qpx_polish_supported(pcode, self.n, self.m, self.q)))
        share_ok = bool(self.lib.dll.qpx_can_share_factors(code, self.n, self.m, self.q))
        self.shared = B > 1 and share_ok and _is_shared(Q, B) and _is_shared(G, B) and _is_shared(A, B)
        nblob = 1 if self.shared else B
        self.sfac = 0 if self.shared else self.elems
        self.blob = torch.empty(nblob * self.elems, dtype=torch.float64 if self.wide else Q.dtype, device=Q.device)
        self.status = torch.empty(B, dtype=torch.int32, device=Q.device)      # every pre-factorisation kernel writes it
        with self._knob():
            self.lib.pre_factor(nblob, self.n, self.m, self.q, Q, G, A if self.q else None, self.blob, self.status,
                                wide=self.wide)
        if self.shared:
            self.status[1:] = self.status[0]
        # The reference raises on a bad Q / A from inside forward (qp.py:81-85, batch.py:379-386).  The two
        # status bits that matter are final once the pre-factorisation kernel has run, so the status words are
        # copied to pinned host memory right behind it and read (raise_on_failure) after the loop kernel
        # has been enqueued: the host waits for the pre-factorisation only, never for the IPM loop.
        self._pre_host = self._pre_event = self._pre_key = self._pre_bits = None
        if self.status.is_cuda:
            # one DMA of the per-QP status words, no reduction kernels in the stream.  (While the stream is being captured
            # into a graph the pool is left alone -- allocating pinned memory or asking an event would invalidate the
            # capture: the words stay on the device and raise_on_failure reads them from there.)
            if not torch.cuda.is_current_stream_capturing():
                self._pre_key, self._pre_host = _PINNED.take(self.device, nblob)
            if self._pre_host is not None:
                self._pre_host.copy_(self.status[:nblob], non_blocking=True)
                self._pre_event = torch.cuda.Event()
                self._pre_event.record(torch.cuda.current_stream(self.device))
        return self

    def _release_pinned(self, done):
        """the pinned buffer goes back to the pool: at once when the copy into it has completed (`done`), else behind its event"""
        host, key = self._pre_host, self._pre_key
        self._pre_host = self._pre_key = None
        if host is not None:
            _PINNED.give(key, host, None if done else self._pre_event)

    def __del__(self):
        try:
            if getattr(self, "_pre_host", None) is not None:
                self._release_pinned(False)
        except Exception:          # interpreter shutdown: nothing to give back to
            pass

    @contextlib.contextmanager
    def _knob(self):
        """around every launch on these factors: the library's A/B knob as it was when they were built, and THEIR
        device current (the launchers opt kernels into > 64 KiB of LDS per device and fork side streams from the
        current device's pool: tensors on cuda:1 while cuda:0 is current must not reach them that way)"""
        dll = self.lib.dll
        old = dll.qpx_set_ipm_variant(self.variant)
        guard = torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()
        try:
            with guard:
                yield
        finally:
            dll.qpx_set_ipm_variant(old)

    # -- error surface of pre_factor_kkt / QPFunction (qp.py:81-85, batch.py:379-386) ------
    def raise_on_failure(self, check_Q_spd=False):
        mask = _lib.ST_Q_NOT_SPD | _lib.ST_A_RANK
        if self._pre_event is not None:
            if self._pre_bits is None:         # first reading: wait for the copy, keep the bits, free the buffer
                self._pre_event.synchronize()
                self._pre_bits = int(np.bitwise_or.reduce(self._pre_host.numpy()))
                self._release_pinned(True)
            st = self._pre_bits & mask
        else:
            st = int(np.bitwise_or.reduce(self.status.cpu().numpy().reshape(-1))) & mask
        if st & _lib.ST_Q_NOT_SPD:
            if check_Q_spd:
                raise RuntimeError('Q is not SPD.')
            raise RuntimeError("""
qpth Error: Cannot perform LU factorization on Q.
Please make sure that your Q matrix is PSD and has
a non-zero diagonal.
""")
        if st & _lib.ST_A_RANK:
            raise RuntimeError("qpth_amd Error: A Q^-1 A^T is not positive definite; "
                               "the equality constraints must have full row rank.")

    def _check(self, X, k, what, batched_ok=True):
        """The kernels index raw pointers: a tensor of another dtype, device or shape must never reach them (the
        reference fails inside bmm / baddbmm with a size or dtype error; here it would be an out-of-bounds read)."""
        if X.dtype != self.dtype or X.device != self.device:
            raise RuntimeError("qpth_amd: %s is %s on %s, the factors were built for %s on %s (all of Q, p, G, h, A, b "
                               "must share one dtype and one device)" % (what, X.dtype, X.device, self.dtype, self.device))
        shape = tuple(X.shape)
        if not (shape == (k,) or (batched_ok and shape in ((self.B, k), (1, k)))):
            raise RuntimeError("qpth_amd: %s has shape %s, expected (%d, %d) or (%d,)" % (what, shape, self.B, k, k))

    def _vec(self, X, k, what="vector"):
        """dense (B,k) contiguous tensor or None"""
        if X is None or X.nelement() == 0 or k == 0:
            return None
        self._check(X, k, what)
        if X.dim() == 1 or X.size(0) != self.B:
            X = X.reshape(1, k).expand(self.B, k)
        return X.contiguous()

    # -- forward (batch.py:47-207) -------------------------------------------------------
    def ipm(self, p, h, b, eps=1e-12, maxIter=20, notImprovedLim=3, stall_policy=None, want_trace=False):
        """The IPM loop (batch.py:47-207), one kernel launch, no host sync.  `result.status` IS the factors'
        status array: the loop ORs its bits (breakdown, maxIter, inaccurate) into the pre-factorisation's,
        so repeated calls on the same factors accumulate them."""
        B, n, m, q = self.B, self.n, self.m, self.q
        dt, dev = self.dtype, self.device
        r = IpmResult()
        r.zhat = torch.empty(B, n, dtype=dt, device=dev)
        r.nu = torch.empty(B, q, dtype=dt, device=dev)
        r.lam = torch.empty(B, m, dtype=dt, device=dev)
        r.slacks = torch.empty(B, m, dtype=dt, device=dev)
        r.iters = torch.empty(B, dtype=torch.int32, device=dev)                # written on every path of the loop kernels
        r.best_resid = torch.empty(B, dtype=dt, device=dev)
        r.trace = torch.full((maxIter, B, 3), float('nan'), dtype=dt, device=dev) if want_trace else None
        r.status = self.status
        if stall_policy is None:
            stall_policy = default_stall_policy(B)
        self._check(p, n, "p")
        self._check(h, m, "h")
        if q:
            if b is None or b.nelement() == 0:
                raise RuntimeError("qpth_amd: A has %d rows but b is empty" % q)
            self._check(b, q, "b")
        with self._knob():
            self.lib.ipm(B, n, m, q, p, h, b if q else None, self.blob, self.sfac, eps, maxIter, notImprovedLim,
                         stall_policy, r.zhat, r.nu if q else None, r.lam, r.slacks, r.iters, self.status,
                         r.best_resid, r.trace, wide=self.wide)
        return r

    # -- factor_kkt + solve_kkt (batch.py:435-470, 349-372) ----------------------------------
    def solve_kkt(self, d, rx, rs, rz, ry, refine=0):
        """factor_kkt + solve_kkt; refine > 0: that many steps of iterative refinement on the residual of the original
        KKT system (batch.py:228-270, KKTSolvers.IR_UNOPT) inside the kernel, re-using the factorisation."""
        B, n, m, q = self.B, self.n, self.m, self.q
        dt, dev = self.dtype, self.device
        d = self._vec(d, m, "d")
        dx = torch.empty(B, n, dtype=dt, device=dev)
        ds = torch.empty(B, m, dtype=dt, device=dev)
        dz = torch.empty(B, m, dtype=dt, device=dev)
        dy = torch.empty(B, q, dtype=dt, device=dev) if q else None
        with self._knob():
            self.lib.factor_solve_kkt(B, n, m, q, self.blob, self.sfac, d, self._vec(rx, n, "rx"), self._vec(rs, m, "rs"),
                                      self._vec(rz, m, "rz"), self._vec(ry, q, "ry"), dx, ds, dz, dy, self.status,
                                      refine=refine, Q=self.Q, G=self.G, A=self.A, wide=self.wide)
        return dx, ds, dz, dy

    # -- KKTSolvers.IR_UNOPT (batch.py:244-270) as a finishing stage --------------------------------
    def polish(self, p, h, b, res, steps=2, refine=0):
        """The finishing stage: `steps` iterations of the reference's PDIPM loop (batch.py:92-198: affine + centring-corrector)
        in the ORIGINAL variables (x, s, z, y), on residuals of the caller's data accumulated in float64, from the loop
        kernel's result; the best iterate is kept -- by the reference's residual ||rx|| + ||rz|| + ||ry|| + nineq mu,
        strict <, NaN never wins (batch.py:118-139) -- so a step that does not help cannot make the answer worse.  The
        loop kernel iterates on pre-computed products (R = G Q^-1 G^T, ...): in float32 their rounding error (cond(Q) ~
        1e6 on the benchmark generator) is a perturbation of the PROBLEM that no number of loop iterations removes;
        residuals against the original data do.  One C call (qpx_polish, include/qpx.h): one kernel where the thread-grid
        / tile kernels serve the size, a stream-ordered sequence of the large-QP family's launches beyond (v7).  No host
        sync.  (Rounds 2-4 composed this stage from torch operations on the host side for the sizes without a kernel;
        that version now lives in tests/polish_reference.py as the step-by-step reference of the kernels.)"""
        if self.polish_ok:
            B, n, m, q = self.B, self.n, self.m, self.q
            self._check(p, n, "p")
            self._check(h, m, "h")
            if q:
                self._check(b, q, "b")
            if not self.refine_ok:
                # the large-QP family has no refinement inside its KKT solves (qpx_refine_supported): the finishing steps
                # themselves run, their solves un-refined -- QPFunction(refine=k) never asks for more
                refine = 0
            for name in ("zhat", "lam", "slacks") + (("nu",) if q else ()):
                setattr(res, name, getattr(res, name).contiguous())
            with self._knob():
                self.lib.polish(B, n, m, q, self.Q, p, self.G, h, self.A if q else None, b if q else None, self.blob, self.sfac,
                                steps, refine, res.zhat, res.nu if q else None, res.lam, res.slacks, None, self.status)
            return res
        if self.wide:
            # float32 tensors in float64 arithmetic: the loop's answer already is the float64 solution of the data (DESIGN
            # 3.4), there is nothing for residuals in float64 to add; rounds 2-4 ran a host-composed stage here
            return res
        raise RuntimeError("qpth_amd: no finishing stage (qpx_polish: KKTSolvers.IR_UNOPT, float32 refine=k) for this size / dtype "
                           "under the current knob -- nz = %d, nineq = %d, neq = %d; it is served up to 512 per dimension "
                           "(qpx_polish_supported).  float32 tensors run in float64 arithmetic by default (refine=None)"
                           % (self.n, self.m, self.q))

    # -- QPFunctionFn.backward (qp.py:127-182) ------------------------------------------------
    def backward(self, zhat, lam, slacks, nu, dl_dz, want=(True,) * 6, shared=(False,) * 6, refine=0):
        """Gradients (dQ, dp, dG, dh, dA, db) for the parameters `want` asks for (ctx.needs_input_grad;
        the others come back as None and cost nothing).  A parameter flagged in `shared` is one the whole
        batch shares: its gradient is returned already reduced to the reference's `.mean(0)` (qp.py:159-177)
        -- for the matrices by one contraction over the batch (qpx_batch_outer) instead of B outer products."""
        B, n, m, q = self.B, self.n, self.m, self.q
        dt, dev = self.dtype, self.device
        wQ, wp, wG, wh, wA, wb = [bool(w) for w in want]
        sQ, sp, sG, sh, sA, sb = [bool(s) for s in shared]
        if q == 0:
            wA = wb = False

        def buf(flag, *shape):
            return torch.empty(*shape, dtype=dt, device=dev) if flag else None

        dQ = buf(wQ and not sQ, B, n, n)
        dG = buf(wG and not sG, B, m, n)
        dA = buf(wA and not sA, B, q, n)
        # per-QP vector gradients come out of the kernel as they are (dp = dx, dh = -dz, db = -dy: qp.py:157-166; the
        # kernel writes the signs, no elementwise launches behind it); the KKT solution itself (dx, dz, dy) is only asked
        # for when a batch-shared parameter needs it: one contraction / mean over the batch
        dp = buf(wp and not sp, B, n)
        dh = buf(wh and not sh, B, m)
        db = buf(wb and not sb, B, q)
        dx = buf((wp and sp) or (wQ and sQ) or (wG and sG) or (wA and sA), B, n)
        dz = buf((wh and sh) or (wG and sG), B, m)
        dy = buf(q > 0 and ((wb and sb) or (wA and sA)), B, q)
        zh, lm, nv = self._vec(zhat, n, "zhat"), self._vec(lam, m, "lam"), self._vec(nu, q, "nu")
        with self._knob():
            self.lib.backward(B, n, m, q, self.blob, self.sfac, zh, lm, self._vec(slacks, m, "slacks"), nv,
                              self._vec(dl_dz, n, "dl_dz"), dQ, dp, dG, dh, dA, db, self.status, dx, dz, dy,
                              refine=refine, Q=self.Q, G=self.G, A=self.A, wide=self.wide)
        with self._knob():           # (their launches too need the factors' device current)
            if wQ and sQ:
                dQ = torch.empty(n, n, dtype=dt, device=dev)
                self.lib.batch_outer(dx, zh, zh, dx, 0.5, dQ)
            if wG and sG:
                dG = torch.empty(m, n, dtype=dt, device=dev)
                self.lib.batch_outer(dz, zh, lm, dx, 1.0, dG)
            if wA and sA:
                dA = torch.empty(q, n, dtype=dt, device=dev)
                self.lib.batch_outer(dy, zh, nv, dx, 1.0, dA)
            # shared vectors: the same contraction with a column of ones (ABI v8) -- `.mean(0)` in a fixed order of
            # additions, the sign of qp.py:160-166 folded into the scale
            if wp and sp:
                dp = torch.empty(n, dtype=dt, device=dev)
                self.lib.batch_outer(dx, None, None, None, 1.0, dp)
            if wh and sh:
                dh = torch.empty(m, dtype=dt, device=dev)
                self.lib.batch_outer(dz, None, None, None, -1.0, dh)
            if wb and sb:
                db = torch.empty(q, dtype=dt, device=dev)
                self.lib.batch_outer(dy, None, None, None, -1.0, db)
        return dQ, dp, dG, dh, dA, db
