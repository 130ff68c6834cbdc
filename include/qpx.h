/*
 * qpx.h -- C ABI of libqpx_hip.so: the MI355X (gfx950) implementation of the one hot path of
 * locuslab/qpth, the dense batched primal-dual interior-point QP solver and its backward pass.
 *
 *     z* = argmin_z 1/2 z'Qz + p'z   s.t.  Gz <= h,  Az = b        (qpth/qp.py:32-42)
 *
 * Every entry point replaces one solver entry point of the reference (the seam that
 * qpth/qp.py:92-96,148-155 calls through):
 *
 *   qpx_pre_factor ........ qpth/solvers/pdipm/batch.py:375-429   pre_factor_kkt(Q, G, A)
 *   qpx_ipm ............... qpth/solvers/pdipm/batch.py:47-207    forward(Q,p,G,h,A,b,Q_LU,S_LU,R,...)
 *   qpx_forward ........... qpth/qp.py:92-96                       pre_factor_kkt + forward
 *   qpx_factor_solve_kkt .. qpth/solvers/pdipm/batch.py:435-470 + 349-372
 *                                                                  factor_kkt(S_LU,R,d); solve_kkt(...)
 *   qpx_backward .......... qpth/qp.py:127-182                     QPFunctionFn.backward (per-QP grads)
 *
 * Conventions
 *   - dtype: QPX_F32 or QPX_F64: every `void*` array below has that element type; or QPX_F32_WIDE (see the enum).
 *   - All pointers are DEVICE pointers valid on `stream` (a hipStream_t).  The caller owns every
 *     buffer; the library never allocates or frees device memory.  Calls are stream-ordered and
 *     re-entrant: the only mutable state is the A/B knob of qpx_set_ipm_variant, which is per host
 *     thread (thread_local) and which nothing but measurements and tests should touch.
 *   - No call synchronises with the host, with ONE exception, in the large-QP family only (nz+neq+nineq > 208) when
 *     a batch of >= 96 QPs is worked on in two parts on two streams (the default there; knob bits 16..19 = 1 turns it
 *     off): the first call of a host thread on a device creates one side stream (kept for the thread's life) and
 *     checks ONCE per (side stream, caller stream) pair -- the last 8 pairs are remembered -- that the two really run
 *     side by side (HIP may deal both to one hardware queue: the parts would serialise, 2 x slower).  That check
 *     enqueues two 100-us delay kernels and waits for them on the host (~0.3 ms; it also waits for whatever the
 *     caller had queued on `stream` before).  It is skipped while `stream` is being captured into a graph and when
 *     the environment variable QPX_NO_STREAM_PROBE is set (then the first side stream is taken as it comes).
 *   - Arrays are dense, row-major, batch-major: Q (B,n,n), p (B,n), G (B,m,n), h (B,m),
 *     A (B,q,n), b (B,q).  A batch stride (in elements) of 0 means "one copy shared by the whole
 *     batch" (the reference's un-batched parameters, qpth/util.py:44-50).  q = 0: A, b unused.
 *   - `factors`: B * qpx_factor_elems(dtype,n,m,q) elements of scratch that carries the
 *     factorisations from qpx_pre_factor to qpx_ipm / qpx_factor_solve_kkt / qpx_backward
 *     (what the reference stashes on ctx as Q_LU, S_LU, R; qpth/qp.py:93).  Consumers take
 *     its batch stride `sfac` in elements: qpx_factor_elems(dtype,n,m,q), or 0 when Q, G, A are
 *     shared by the whole batch and were factored once with B = 1 (only if qpx_can_share_factors()).
 *   - `status`: int32[B], per-QP bit mask of QPX_ST_* written on the device; the functions
 *     themselves return 0 or a negative QPX_ERR_* launch/argument error and never throw.
 *   - Semantics of the IPM loop are those of the reference for a batch of one per QP; see
 *     `stall_policy` and DESIGN.md for the batch-global quirks of the reference this removes.
 */
#ifndef QPX_H
#define QPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QPX_ABI_VERSION 8

/* QPX_F32_WIDE (ABI v4): the caller's arrays are float32, `factors` and all arithmetic are float64 -- every `void*`
 * array below except `factors` has float elements, `factors` holds qpx_factor_elems(QPX_F32_WIDE, ...) DOUBLES.  On
 * MI355X the float64 matrix-core kernels are as fast as the float32 thread-grid kernels and return the float64
 * solution of the float32 data (the float32 kernels iterate on products rounded to float32: ~1e-4 from it on the
 * reference's benchmark generator).  Served where the thread-grid / tile kernels run (nz+neq+nineq <= 208) and (v6) by
 * the large-QP family, else QPX_ERR_UNSUPPORTED; refine must be 0; qpx_batch_outer takes QPX_F32 for such a caller's
 * float32 vectors. */
enum { QPX_F32 = 0, QPX_F64 = 1, QPX_F32_WIDE = 2 };

enum {
    QPX_OK = 0,
    QPX_ERR_ARG = -1,          /* bad dtype / sizes / null pointer            */
    QPX_ERR_UNSUPPORTED = -2,  /* max(n,m,q) beyond what this build dispatches */
    QPX_ERR_LAUNCH = -3,       /* HIP launch error (see hipGetLastError)       */
    QPX_ERR_NO_DEVICE = -4
};

/* per-QP status bits */
enum {
    QPX_ST_Q_NOT_SPD = 1,      /* -> RuntimeError('Q is not SPD.')       qp.py:85          */
    QPX_ST_A_RANK = 2,         /* A Q^-1 A^T not factorable              batch.py:407       */
    QPX_ST_KKT_BREAKDOWN = 4,  /* factor_kkt failed, best iterate returned  batch.py:110-113 */
    QPX_ST_INACCURATE = 8,     /* best residual > 1 -> INACC_ERR warning   batch.py:141,205  */
    QPX_ST_MAXITER = 16,
    QPX_ST_NONFINITE = 32
};

/* stall_policy of qpx_ipm: how `notImprovedLim` (batch.py:127-140, batch-global in the
 * reference) is applied per QP.  0 = never stop on stall; 1 = the reference's counter per QP
 * (identical to the reference for a batch of one); 2 = round-off-floor rule (default for B > 1). */
enum { QPX_STALL_OFF = 0, QPX_STALL_REFERENCE = 1, QPX_STALL_FLOOR = 2 };

typedef void* qpx_stream_t; /* hipStream_t */

int qpx_abi_version(void);
const char* qpx_strerror(int code);

/* elements (of dtype) of factor storage per QP (C2, f64: 27 100 = 217 KB; B = 65536, n = m = 64: 5.5 GB) */
size_t qpx_factor_elems(int dtype, int n, int m, int q);

/* largest max(n,m,q) this build can solve (1 024 since round 6; qpx_polish: 512, see qpx_polish_supported) */
int qpx_max_dim(void);
/* QPX_OK if (dtype, n, m, q) is served under the calling thread's knob, else the QPX_ERR_* the entry points would
 * return (QPX_F32_WIDE: QPX_ERR_UNSUPPORTED outside the thread-grid / tile kernels' sizes) */
int qpx_supported(int dtype, int n, int m, int q);
/* v6: which kernel family serves (dtype, n, m, q) under the calling thread's knob (or a negative QPX_ERR_*):
 * the thread-grid kernels, the float64 matrix-core tile kernels (nineq <= 112, nz+neq+nineq <= 208), or the large-QP
 * family (multi-kernel blocked Cholesky / GEMM path).  The host mirror uses it to decide where float32 tensors run in
 * float64 arithmetic (QPX_F32_WIDE: the last two).  (0 was the family of the round-1 workgroup kernels, deleted in v7.) */
enum { QPX_FAMILY_GRID = 1, QPX_FAMILY_TILE = 2, QPX_FAMILY_BIG = 3 };
int qpx_kernel_family(int dtype, int n, int m, int q);
/* v5: 1 if qpx_factor_solve_kkt / qpx_backward implement refine > 0 for this size and dtype (else they return
 * QPX_ERR_UNSUPPORTED when asked to): the in-kernel iterative refinement of KKTSolvers.IR_UNOPT, batch.py:244-270. */
int qpx_refine_supported(int dtype, int n, int m, int q);

/* tuning/A-B knob (per host thread): which kernel family runs.  0 (default) = automatic: the thread-grid /
 * matrix-core kernels (sweep pre-factorisation, register-resident LDL^T with in-place inverse factor)
 * whenever nz+neq+nineq <= 208, else the large-QP family (batched multi-kernel blocked Cholesky / MFMA GEMM path,
 * qpx_big.h -- BASELINE.json configs[3], nz = nineq = 500; equality constraints included since v6);
 * 3 = always the large-QP family.  (1 selected the round-1 workgroup kernels until v7: deleted, the value now means 0.)
 * Adding 256 / 512 / 1024 forces the 16x16-thread grid / the 8x8-thread grid / the matrix-core tile form
 * (f64, nineq <= 112) of the loop kernel, adding 2048 / 4096 / 8192 fixes the tile form's waves per QP
 * at 1 / 2 / 4 (four waves at 4 or 7 tile rows = the chain-wave form, which is the default there); by default the
 * library picks by dtype and size.  Bit 14 (16384): qpx_pre_factor by the symmetric sweep on the thread grid also
 * where its matrix-core form serves the size (float64 arithmetic, 33 <= nz + neq <= 112, nineq <= 112: a
 * factorisation of Q -- of [[Q, A^T], [A, 0]] -- + tile products, qpx_prefac.h; same blob).  (Bit 15, and bit 14 until v6, selected two round-3
 * forms -- a pre-factorisation by a sweep on matrix-core tiles and the four-wave tile kernels without their chain wave
 * -- that lost their A/Bs and were deleted.)
 * Large-QP family only: bits 16..19 = number of parts (1..4) the batch is split into, each part enqueued on a
 * stream of its own (the caller's + side streams forked from and joined back into it with events; the one host wait
 * this can involve is described under "Conventions"), 0 = automatic: two parts from 96 QPs up, else one, in
 * qpx_pre_factor, qpx_ipm / qpx_forward and qpx_polish; one part in qpx_factor_solve_kkt / qpx_backward (one factorisation +
 * one solve is too short a sequence to gain: profiles/r06q_backward_parts.txt), which take parts only from an explicit value; bits 20..24 = initial stagger between the side streams in units of 16 us; (bit 25 selected
 * the four-wave substitutions of round 3 until v7: retired with them); bit 26 = the mat-vec R z' in front of the factorisation
 * in the caller's stream (the round-3 order) instead of beside it on a helper stream; bit 27 =
 * diagonal blocks eliminated by one wave (the round-3 form) instead of four in the chain-wave form; bit 28 = on the
 * thread grid; bit 29 = no XCD-aware tile order.
 * The knob must not change between qpx_pre_factor and the calls that consume its factors (it selects the
 * layout of `factors` too: ask qpx_factor_elems after setting it).
 * Bits and values the library does not decode (retired knobs) are dropped, not stored: qpx_get_ipm_variant returns what
 * was understood, so set-then-get tells a script that its knob no longer exists.
 * Returns the previous value. */
int qpx_set_ipm_variant(int variant);
int qpx_get_ipm_variant(void);      /* the calling thread's current value */

/* May a batch whose Q, G, A are shared be served by ONE factor blob (qpx_pre_factor with B = 1, consumers
 * with sfac = 0)?  Yes for the thread-grid / tile kernels, which only read the blob; no for the large-QP family, which keeps
 * per-QP work matrices in it. */
int qpx_can_share_factors(int dtype, int n, int m, int q);

/* Measurement hook (bench.py): re-issues the largest GEMM of the large-QP pre-factorisation, R = Zt Zt^T, on the
 * blobs qpx_pre_factor wrote (idempotent: R is rewritten with the same values), as ONE launch on `stream`, so that
 * it can be bracketed by events on its own.  QPX_ERR_UNSUPPORTED when (dtype, n, m, q) is not served by that family. */
int qpx_big_gemm_r(int dtype, int B, int n, int m, int q, void* factors, qpx_stream_t stream);

/* pre_factor_kkt(Q, G, A) */
int qpx_pre_factor(int dtype, int B, int n, int m, int q,
                   const void* Q, int64_t sQ, const void* G, int64_t sG, const void* A, int64_t sA,
                   void* factors, int32_t* status, qpx_stream_t stream);

/* forward(Q,p,G,h,A,b,Q_LU,S_LU,R,...): the PDIPM loop on pre-factored QPs.  Outputs in the
 * reference's return order x, y, z, s = zhat (B,n), nu (B,q), lam (B,m), slacks (B,m);
 * iters int32[B]; best_resid dtype[B]; trace: NULL or dtype[maxIter][B][3] = (pri_resid,
 * dual_resid, mu) per iteration (what verbose=1 prints, batch.py:115-117). */
int qpx_ipm(int dtype, int B, int n, int m, int q,
            const void* p, int64_t sp, const void* h, int64_t sh, const void* b, int64_t sb,
            void* factors, int64_t sfac, double eps, int maxIter, int notImprovedLim, int stall_policy,
            void* zhat, void* nu, void* lam, void* slack,
            int32_t* iters, int32_t* status, void* best_resid, void* trace, qpx_stream_t stream);

/* qpx_pre_factor followed by qpx_ipm on the same stream */
int qpx_forward(int dtype, int B, int n, int m, int q,
                const void* Q, int64_t sQ, const void* p, int64_t sp,
                const void* G, int64_t sG, const void* h, int64_t sh,
                const void* A, int64_t sA, const void* b, int64_t sb,
                void* factors, double eps, int maxIter, int notImprovedLim, int stall_policy,
                void* zhat, void* nu, void* lam, void* slack,
                int32_t* iters, int32_t* status, void* best_resid, void* trace, qpx_stream_t stream);

/* factor_kkt(S_LU, R, d) then solve_kkt(Q_LU, d, G, A, S_LU, rx, rs, rz, ry) -> dx, ds, dz, dy.
 * d (B,m) > 0; rx (B,n), rs, rz (B,m), ry (B,q): NULL means zeros; dy may be NULL when q = 0.
 * refine > 0: that many steps of ITERATIVE REFINEMENT on the residual of the original KKT system
 * (qpth/solvers/pdipm/batch.py:228-270, kkt_resid_reg + solve_kkt_ir -- KKTSolvers.IR_UNOPT), evaluated with the
 * caller's Q (sQ), G (sG), A (sA) (batch strides in elements, 0 = shared); the factorisation is re-used, not repeated.
 * Implemented by the thread-grid / tile kernels (nz+neq+nineq <= 208, dtype QPX_F32 / QPX_F64): qpx_refine_supported.
 * Anywhere else refine > 0 returns QPX_ERR_UNSUPPORTED (v5; up to v4 the other kernel families ignored it). */
int qpx_factor_solve_kkt(int dtype, int B, int n, int m, int q, void* factors, int64_t sfac,
                         const void* d, const void* rx, const void* rs, const void* rz, const void* ry,
                         void* dx, void* ds, void* dz, void* dy,
                         int refine, const void* Q, int64_t sQ, const void* G, int64_t sG, const void* A, int64_t sA,
                         int32_t* status, qpx_stream_t stream);

/* QPFunctionFn.backward for given (zhat, lam, slacks, nu) -- from qpx_ipm or from any other
 * solver (qp.py:142-155) -- and dl_dz (B,n).  Per-QP gradients dQ (B,n,n), dp (B,n),
 * dG (B,m,n), dh (B,m), dA (B,q,n), db (B,q); each of the six may be NULL = "this gradient is not
 * wanted" (ctx.needs_input_grad): nothing is computed or written for it.  dx (B,n), dz (B,m),
 * dy (B,q): optional (NULL = skip) solution of the backward KKT system itself (qp.py:151-155), the
 * inputs of qpx_batch_outer.  refine, Q, G, A: as for qpx_factor_solve_kkt (0 / NULL: no refinement). */
int qpx_backward(int dtype, int B, int n, int m, int q, void* factors, int64_t sfac,
                 const void* zhat, const void* lam, const void* slack, const void* nu, const void* dl_dz,
                 void* dQ, void* dp, void* dG, void* dh, void* dA, void* db,
                 void* dx, void* dz, void* dy,
                 int refine, const void* Q, int64_t sQ, const void* G, int64_t sG, const void* A, int64_t sA,
                 int32_t* status, qpx_stream_t stream);

/* v6: the FINISHING STAGE as one kernel -- `steps` iterations of the reference's PDIPM loop in the original variables
 * (qpth/solvers/pdipm/batch.py:92-198: affine + centring-corrector Newton steps, step lengths batch.py:189-198) started
 * from the iterate (zhat, nu, lam, slack) the caller passes in, with the KKT residuals (batch.py:93-101) formed from the
 * caller's Q, p, G, h, A, b in float64 accumulation whatever dtype is, every KKT solve through the factors of
 * qpx_pre_factor and refined `refine` times on the residual of the original system (kkt_resid_reg / solve_kkt_ir,
 * batch.py:228-270 -- what forward(solver=KKTSolvers.IR_UNOPT) asks for).  The four arrays are overwritten with the BEST
 * iterate met (the reference's rule, batch.py:118-139: residual ||rx|| + ||rz|| + ||ry|| + nineq mu, strict <, NaN never
 * wins; the start iterate competes), best_resid (dtype[B], may be NULL) with its residual.  Served by every kernel
 * family since v7 (dtype QPX_F32 or QPX_F64; one kernel up to nz+neq+nineq = 208, the large-QP family's launch sequence
 * beyond, where `refine` must be 0: qpx_refine_supported, and up to max(n,m,q) = 512 only: its vectors live in LDS);
 * qpx_polish_supported says so; QPX_F32_WIDE and sizes beyond that: QPX_ERR_UNSUPPORTED.  Strides as everywhere: elements, 0 = shared by the batch. */
int qpx_polish_supported(int dtype, int n, int m, int q);
int qpx_polish(int dtype, int B, int n, int m, int q,
               const void* Q, int64_t sQ, const void* p, int64_t sp, const void* G, int64_t sG, const void* h, int64_t sh,
               const void* A, int64_t sA, const void* b, int64_t sb,
               void* factors, int64_t sfac, int steps, int refine,
               void* zhat, void* nu, void* lam, void* slack, void* best_resid,
               int32_t* status, qpx_stream_t stream);

/* Batch-MEAN of the gradient of a parameter that the whole batch shares (qp.py:159-177: the reference
 * forms B outer products and then `.mean(0)`): one contraction over the batch instead,
 *   out (r,c) = scale/B * sum_b ( u[b][r] v[b][c] + w[b][r] x[b][c] )        u, w: (B,r)  v, x: (B,c)
 * dQ: u = dx, v = zhat, w = zhat, x = dx, scale = 0.5;  dG: u = dz, v = zhat, w = lam, x = dx, scale = 1;
 * dA: u = dy, v = zhat, w = nu, x = dx.  `out` is (r,c) of dtype, overwritten.
 * v8: w == x == NULL: the second product is left out; v == NULL (c must be 1): v is a column of ones, i.e.
 *   out (r) = scale/B * sum_b u[b][r]  -- the `.mean(0)` of a shared VECTOR parameter's gradient (dp = mean dx,
 *   dh = -mean dz, db = -mean dy: qp.py:160-166,174-177), in the same fixed order of additions. */
int qpx_batch_outer(int dtype, int B, int r, int c, const void* u, const void* v, const void* w,
                    const void* x, double scale, void* out, void* ws, size_t ws_elems, qpx_stream_t stream);
/* v7: `ws` -- ws_elems elements of dtype, at least qpx_batch_outer_workspace_elems(dtype, B, r, c) -- lets a long batch be
 * contracted in TWO stages: partial tiles per chunk of the batch by many workgroups, then their sum in chunk order (fixed
 * order: bit-reproducible, no atomics).  NULL / too small / 0 elements needed: one workgroup per 16 x 16 tile of `out`
 * walks the whole batch, as before -- the same result up to the order of the additions. */
size_t qpx_batch_outer_workspace_elems(int dtype, int B, int r, int c);

/* v7: x = M^-1 r for B general k x k systems (Gaussian elimination with partial pivoting, one workgroup each).  M (B,k,k)
 * is destroyed, rhs (B,k) is overwritten by the solution; a singular M ORs QPX_ST_KKT_BREAKDOWN into status[b] (may be
 * NULL) and returns NaNs.  The host mirror's factor_solve_kkt_reg (qpth/solvers/pdipm/batch.py:273-310) with equality
 * constraints uses it for the neq x neq correction of the regularised (y, y) block. */
int qpx_dense_solve(int dtype, int B, int k, void* M, void* rhs, int32_t* status, qpx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QPX_H */
