/*
 * oracle/qp_oracle.c -- CPU restatement of the reference's dense batched PDIPM path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker* for the HIP path: it may be
 * called from tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg and
 * from nowhere else.  Nothing under qpth_amd/ imports, links or executes it.
 *
 * It restates, in plain C, the algorithm of the reference (locuslab/qpth v0.0.18):
 *
 *   lu_factor / lu_solve .......... the ATen->LAPACK getrf/getrs calls made by
 *                                   qpth/solvers/pdipm/batch.py:9 (lu_hack, CPU branch:
 *                                   partial pivoting) and batch.py:353,360,367
 *   qpo_pre_factor ................ qpth/solvers/pdipm/batch.py:375-429 (pre_factor_kkt)
 *   qpo_factor_kkt ................ qpth/solvers/pdipm/batch.py:435-470 (factor_kkt)
 *   qpo_solve_kkt ................. qpth/solvers/pdipm/batch.py:349-372 (solve_kkt)
 *   get_step ...................... qpth/solvers/pdipm/batch.py:210-213
 *   qpo_forward ................... qpth/solvers/pdipm/batch.py:47-207 (forward) driven as
 *                                   qpth/qp.py:92-96 drives it
 *   qpo_backward .................. qpth/qp.py:127-182 (QPFunctionFn.backward), per-QP
 *                                   gradients before the broadcast `.mean(0)`
 *
 * The arithmetic of the reference lives in a third-party dependency that is not under
 * /root/reference: PyTorch ATen -> LAPACK/MKL getrf/getrs (un-pinned; setup.py:13-16 does
 * not even list torch).  getrf's published algorithm (right-looking LU with partial
 * pivoting, row interchanges recorded 1-based... here 0-based) is restated in lu_factor().
 * Parity is pinned on outputs of the reference itself run in the build container
 * (tests/golden/make_golden.py -> tests/golden/ *.npz); see tests/test_oracle_golden.py.
 *
 * Two termination modes:
 *   per_qp = 0  reference semantics exactly: `nNotImproved`, `best.resids.max() < eps`,
 *               `mu.min() > 1e32` and get_step's `a.max()` are batch-global
 *               (batch.py:127-141, 212).
 *   per_qp = 1  every QP is treated as the reference treats a batch of one (this is what a
 *               one-QP-per-workgroup kernel can implement); `stall_policy` selects how the
 *               not-improved counter is applied per QP:
 *                 0 never stop on stall, 1 reference counter per QP (== reference at B=1),
 *                 2 "round-off floor" rule (the default of the HIP path for B > 1): in exact
 *                   arithmetic the feasibility residual obeys feas_{k+1} = (1-alpha_k) feas_k;
 *                   once the measured one exceeds twice that prediction it is round-off, and
 *                   the QP stops as soon as nineq*mu < 1e-2 * feas.  The not-improved counter
 *                   only counts while nineq*mu < feas.
 *
 * The source is compiled twice (REAL=double / REAL=float); symbols carry SUFFIX.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#define SUFFIX _f64
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef REAL real;

/* ---- LAPACK-style kernels (batch.py:9 lu_hack CPU branch -> getrf; lu_solve -> getrs) ---- */

/* In-place LU with partial pivoting of the row-major n x n matrix a (leading dim n).
 * piv[k] = row swapped with k at step k.  Returns 0, or k+1 if U[k][k] is exactly zero /
 * NaN-free check is left to the caller (LAPACK semantics). */
static int lu_factor(int n, real *a, int *piv)
{
    int info = 0;
    for (int k = 0; k < n; ++k) {
        int p = k;
        real best = fabs((double)a[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            real v = fabs((double)a[i * n + k]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < n; ++j) {
                real t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t;
            }
        real d = a[k * n + k];
        if (d == (real)0) { if (!info) info = k + 1; continue; }
        real inv = (real)1 / d;
        for (int i = k + 1; i < n; ++i) {
            real l = a[i * n + k] * inv;
            a[i * n + k] = l;
            if (l != (real)0) {
                const real *rk = a + k * n;
                real *ri = a + i * n;
                for (int j = k + 1; j < n; ++j) ri[j] -= l * rk[j];
            }
        }
    }
    return info;
}

/* Solve (P L U) x = b in place for nrhs right-hand sides stored as b[i*ldb + r]. */
static void lu_solve(int n, const real *lu, const int *piv, real *b, int nrhs, int ldb)
{
    for (int k = 0; k < n; ++k) {
        int p = piv[k];
        if (p != k)
            for (int r = 0; r < nrhs; ++r) {
                real t = b[k * ldb + r]; b[k * ldb + r] = b[p * ldb + r]; b[p * ldb + r] = t;
            }
    }
    for (int i = 1; i < n; ++i)
        for (int k = 0; k < i; ++k) {
            real l = lu[i * n + k];
            if (l != (real)0)
                for (int r = 0; r < nrhs; ++r) b[i * ldb + r] -= l * b[k * ldb + r];
        }
    for (int i = n - 1; i >= 0; --i) {
        for (int k = i + 1; k < n; ++k) {
            real u = lu[i * n + k];
            if (u != (real)0)
                for (int r = 0; r < nrhs; ++r) b[i * ldb + r] -= u * b[k * ldb + r];
        }
        real inv = (real)1 / lu[i * n + i];
        for (int r = 0; r < nrhs; ++r) b[i * ldb + r] *= inv;
    }
}

/* ---- per-QP factor state (what the reference keeps in Q_LU, S_LU, R) ---- */
typedef struct {
    int n, m, q;
    real *Q_lu;     int *Q_piv;        /* LU(Q)                       batch.py:380          */
    real *invQ_GT;                     /* Q^-1 G^T   (n x m)          batch.py:398          */
    real *R;                           /* Schur block (m x m)         batch.py:396-399,424  */
    real *S11_lu;   int *S11_piv;      /* LU(A Q^-1 A^T) (q x q)      batch.py:403-408      */
    real *S21;                         /* G Q^-1 A^T (m x q)          batch.py:405          */
    real *T_lu;     int *T_piv;        /* LU(R + diag(1/d)) (m x m)   batch.py:445-448      */
} qp_factors;

static size_t fac_reals(int n, int m, int q)
{
    return (size_t)n * n + (size_t)n * m + (size_t)m * m + (size_t)q * q + (size_t)m * q +
           (size_t)m * m;
}
static size_t fac_ints(int n, int m, int q) { return (size_t)n + q + m; }

static void fac_bind(qp_factors *f, int n, int m, int q, real *rbuf, int *ibuf)
{
    f->n = n; f->m = m; f->q = q;
    f->Q_lu = rbuf;            rbuf += (size_t)n * n;
    f->invQ_GT = rbuf;         rbuf += (size_t)n * m;
    f->R = rbuf;               rbuf += (size_t)m * m;
    f->S11_lu = rbuf;          rbuf += (size_t)q * q;
    f->S21 = rbuf;             rbuf += (size_t)m * q;
    f->T_lu = rbuf;
    f->Q_piv = ibuf;           ibuf += n;
    f->S11_piv = ibuf;         ibuf += q;
    f->T_piv = ibuf;
}

/* pre_factor_kkt (batch.py:375-429) for one QP.  Returns 0 or an error code:
 * 1 = LU(Q) hit a zero pivot (the reference raises, batch.py:381-386), 2 = LU(A Q^-1 A^T). */
static int pre_factor_one(qp_factors *f, const real *Q, const real *G, const real *A)
{
    const int n = f->n, m = f->m, q = f->q;
    memcpy(f->Q_lu, Q, sizeof(real) * n * n);
    if (lu_factor(n, f->Q_lu, f->Q_piv)) return 1;
    /* invQ_GT = Q^-1 G^T : rhs matrix is n x m with rhs index r = constraint */
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < m; ++r) f->invQ_GT[i * m + r] = G[r * n + i];
    lu_solve(n, f->Q_lu, f->Q_piv, f->invQ_GT, m, m);
    /* R = G invQ_GT  (batch.py:396-399) */
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            real acc = 0;
            for (int k = 0; k < n; ++k) acc += G[i * n + k] * f->invQ_GT[k * m + j];
            f->R[i * m + j] = acc;
        }
    if (q > 0) {
        /* invQ_AT (n x q), A_invQ_AT (q x q), G_invQ_AT (m x q)   batch.py:403-405 */
        real *invQ_AT = (real *)malloc(sizeof(real) * n * q);
        real *Tm = (real *)malloc(sizeof(real) * q * m);
        for (int i = 0; i < n; ++i)
            for (int r = 0; r < q; ++r) invQ_AT[i * q + r] = A[r * n + i];
        lu_solve(n, f->Q_lu, f->Q_piv, invQ_AT, q, q);
        for (int i = 0; i < q; ++i)
            for (int j = 0; j < q; ++j) {
                real acc = 0;
                for (int k = 0; k < n; ++k) acc += A[i * n + k] * invQ_AT[k * q + j];
                f->S11_lu[i * q + j] = acc;
            }
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < q; ++j) {
                real acc = 0;
                for (int k = 0; k < n; ++k) acc += G[i * n + k] * invQ_AT[k * q + j];
                f->S21[i * q + j] = acc;
            }
        int info = lu_factor(q, f->S11_lu, f->S11_piv);     /* batch.py:407 */
        if (info) { free(invQ_AT); free(Tm); return 2; }
        /* T = (A Q^-1 A^T)^-1 (G Q^-1 A^T)^T (q x m); R -= G_invQ_AT T   batch.py:415-424 */
        for (int i = 0; i < q; ++i)
            for (int j = 0; j < m; ++j) Tm[i * m + j] = f->S21[j * q + i];
        lu_solve(q, f->S11_lu, f->S11_piv, Tm, m, m);
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                real acc = 0;
                for (int k = 0; k < q; ++k) acc += f->S21[i * q + k] * Tm[k * m + j];
                f->R[i * m + j] -= acc;
            }
        free(invQ_AT); free(Tm);
    }
    return 0;
}

/* factor_kkt (batch.py:435-470): T = R + diag(1/d), LU(T).  Returns LAPACK-style info. */
static int factor_kkt_one(qp_factors *f, const real *d)
{
    const int m = f->m;
    memcpy(f->T_lu, f->R, sizeof(real) * m * m);
    for (int i = 0; i < m; ++i) f->T_lu[i * m + i] += (real)1 / d[i];
    /* NaN/Inf input does not raise in the reference (torch.linalg.lu_factor only raises on
     * an exactly-zero pivot); NaNs simply propagate and never win the best-iterate test. */
    return lu_factor(m, f->T_lu, f->T_piv);
}

/* solve_kkt (batch.py:349-372).  ry/dy may be NULL when q == 0.  Block elimination of
 * S = [[S11, S21^T], [S21, R0 + D^-1]] through its Schur complement T = R + D^-1, which is
 * what the reference's partial block-LU of S (batch.py:388-424, 445-470) evaluates. */
static void solve_kkt_one(const qp_factors *f, const real *G, const real *A, const real *d,
                          const real *rx, const real *rs, const real *rz, const real *ry,
                          real *dx, real *ds, real *dz, real *dy, real *work /* n+m+2q */)
{
    const int n = f->n, m = f->m, q = f->q;
    real *u = work, *h2 = work + n, *h1 = h2 + m, *t1 = h1 + q;
    for (int i = 0; i < n; ++i) u[i] = rx[i];
    lu_solve(n, f->Q_lu, f->Q_piv, u, 1, 1);                      /* invQ_rx  batch.py:353 */
    for (int i = 0; i < q; ++i) {                                  /* batch.py:355 */
        real acc = 0;
        for (int k = 0; k < n; ++k) acc += A[i * n + k] * u[k];
        h1[i] = acc - ry[i];
    }
    for (int i = 0; i < m; ++i) {                                  /* batch.py:356-358 */
        real acc = 0;
        for (int k = 0; k < n; ++k) acc += G[i * n + k] * u[k];
        h2[i] = acc + rs[i] / d[i] - rz[i];
    }
    /* w = -S^-1 h   (batch.py:360) */
    for (int i = 0; i < q; ++i) t1[i] = -h1[i];
    if (q > 0) lu_solve(q, f->S11_lu, f->S11_piv, t1, 1, 1);       /* S11^-1 (-h1) */
    for (int i = 0; i < m; ++i) {
        real acc = -h2[i];
        for (int k = 0; k < q; ++k) acc -= f->S21[i * q + k] * t1[k];
        dz[i] = acc;
    }
    lu_solve(m, f->T_lu, f->T_piv, dz, 1, 1);                      /* w2 */
    for (int i = 0; i < q; ++i) {
        real acc = -h1[i];
        for (int k = 0; k < m; ++k) acc -= f->S21[k * q + i] * dz[k];
        dy[i] = acc;
    }
    if (q > 0) lu_solve(q, f->S11_lu, f->S11_piv, dy, 1, 1);       /* w1 */
    /* g1 = -rx - G^T w2 - A^T w1 ; dx = Q^-1 g1   (batch.py:362-367) */
    for (int i = 0; i < n; ++i) {
        real acc = -rx[i];
        for (int k = 0; k < m; ++k) acc -= G[k * n + i] * dz[k];
        for (int k = 0; k < q; ++k) acc -= A[k * n + i] * dy[k];
        dx[i] = acc;
    }
    lu_solve(n, f->Q_lu, f->Q_piv, dx, 1, 1);
    for (int i = 0; i < m; ++i) ds[i] = (-rs[i] - dz[i]) / d[i];   /* batch.py:365,368 */
}

/* torch semantics helpers: min/max that propagate NaN (torch.min / torch.max / Tensor.min). */
static real nan_min(real a, real b) { return (isnan((double)a) || isnan((double)b)) ? (real)NAN : (a < b ? a : b); }
static real nan_max(real a, real b) { return (isnan((double)a) || isnan((double)b)) ? (real)NAN : (a > b ? a : b); }

/* get_step (batch.py:210-213) for one QP given the replacement value `fill` used for
 * entries with dv > 0 (the reference's `max(1.0, a.max())`, a.max() being batch-global). */
static real get_step_row(int m, const real *v, const real *dv, real fill)
{
    real mn = (real)INFINITY;
    for (int i = 0; i < m; ++i) {
        real a = -v[i] / dv[i];
        if (dv[i] > 0) a = fill;
        mn = nan_min(mn, a);
    }
    return mn;
}
/* max over one row of a = -v/dv (all entries, before masking), NaN-propagating. */
static real get_step_rowmax(int m, const real *v, const real *dv)
{
    real mx = -(real)INFINITY;
    for (int i = 0; i < m; ++i) mx = nan_max(mx, -v[i] / dv[i]);
    return mx;
}
/* python `max(1.0, t)`: returns t only if t > 1.0 (NaN -> 1.0) */
static real py_max1(real t) { return (t > (real)1) ? t : (real)1; }

/* ------------------------------------------------------------------------------------ */
/* Public entry points (ctypes).  All arrays are dense, batch-major, row-major.           */

/* bytes a caller must provide for the factor state of a batch */
size_t FN(qpo_state_bytes)(int B, int n, int m, int q)
{
    return (size_t)B * (fac_reals(n, m, q) * sizeof(real) + fac_ints(n, m, q) * sizeof(int));
}

static void bind_batch(qp_factors *F, void *state, int B, int n, int m, int q)
{
    real *rb = (real *)state;
    int *ib = (int *)((char *)state + (size_t)B * fac_reals(n, m, q) * sizeof(real));
    for (int i = 0; i < B; ++i)
        fac_bind(&F[i], n, m, q, rb + (size_t)i * fac_reals(n, m, q), ib + (size_t)i * fac_ints(n, m, q));
}

/* pre_factor_kkt for a batch; err[i] per QP; returns number of failed QPs */
int FN(qpo_pre_factor)(int B, int n, int m, int q, const real *Q, const real *G, const real *A,
                       void *state, int *err, int nthreads)
{
    qp_factors *F = (qp_factors *)malloc(sizeof(qp_factors) * B);
    bind_batch(F, state, B, n, m, q);
    int nfail = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) reduction(+ : nfail)
    for (int i = 0; i < B; ++i) {
        int e = pre_factor_one(&F[i], Q + (size_t)i * n * n, G + (size_t)i * m * n,
                               q > 0 ? A + (size_t)i * q * n : NULL);
        if (err) err[i] = e;
        nfail += e != 0;
    }
    free(F);
    return nfail;
}

/* factor_kkt for a batch (batch.py:435-470) */
int FN(qpo_factor_kkt)(int B, int n, int m, int q, void *state, const real *d, int nthreads)
{
    qp_factors *F = (qp_factors *)malloc(sizeof(qp_factors) * B);
    bind_batch(F, state, B, n, m, q);
    int nfail = 0;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) reduction(+ : nfail)
    for (int i = 0; i < B; ++i) nfail += factor_kkt_one(&F[i], d + (size_t)i * m) != 0;
    free(F);
    return nfail;
}

/* solve_kkt for a batch (batch.py:349-372) */
void FN(qpo_solve_kkt)(int B, int n, int m, int q, const void *state, const real *G, const real *A,
                       const real *d, const real *rx, const real *rs, const real *rz, const real *ry,
                       real *dx, real *ds, real *dz, real *dy, int nthreads)
{
    qp_factors *F = (qp_factors *)malloc(sizeof(qp_factors) * B);
    bind_batch(F, (void *)state, B, n, m, q);
#pragma omp parallel num_threads(nthreads)
    {
        real *work = (real *)malloc(sizeof(real) * (n + m + 2 * q + 4));
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < B; ++i)
            solve_kkt_one(&F[i], G + (size_t)i * m * n, q > 0 ? A + (size_t)i * q * n : NULL,
                          d + (size_t)i * m, rx + (size_t)i * n, rs + (size_t)i * m, rz + (size_t)i * m,
                          q > 0 ? ry + (size_t)i * q : NULL, dx + (size_t)i * n, ds + (size_t)i * m,
                          dz + (size_t)i * m, q > 0 ? dy + (size_t)i * q : NULL, work);
        free(work);
    }
    free(F);
}

/*
 * forward (batch.py:47-207) on a pre-factored batch.  Outputs: x (B,n), y (B,q), z (B,m),
 * s (B,m) = best iterate per QP (return order of batch.py:143,207), iters[B] = IPM
 * iterations each QP took part in, best_resid[B].  trace (optional, maxIter*3 reals):
 * batch means of pri_resid, dual_resid, mu per iteration exactly as verbose=1 prints them
 * (batch.py:115-117).  Returns the number of loop trips executed.
 */
int FN(qpo_forward)(int B, int n, int m, int q, const real *Q, const real *p, const real *G,
                    const real *h, const real *A, const real *b, void *state, double eps,
                    int maxIter, int notImprovedLim, int per_qp, int stall_policy, real *x_out,
                    real *y_out, real *z_out, real *s_out, int *iters, real *best_resid,
                    real *trace, int nthreads)
{
    qp_factors *F = (qp_factors *)malloc(sizeof(qp_factors) * B);
    bind_batch(F, state, B, n, m, q);
    const size_t nv = (size_t)B * n, mv = (size_t)B * m, qv = (size_t)B * (q > 0 ? q : 1);
    real *x = (real *)calloc(nv, sizeof(real)), *s = (real *)calloc(mv, sizeof(real));
    real *z = (real *)calloc(mv, sizeof(real)), *y = (real *)calloc(qv, sizeof(real));
    real *d = (real *)calloc(mv, sizeof(real));
    real *rx = (real *)calloc(nv, sizeof(real)), *rs = (real *)calloc(mv, sizeof(real));
    real *rz = (real *)calloc(mv, sizeof(real)), *ry = (real *)calloc(qv, sizeof(real));
    real *dxa = (real *)calloc(nv, sizeof(real)), *dsa = (real *)calloc(mv, sizeof(real));
    real *dza = (real *)calloc(mv, sizeof(real)), *dya = (real *)calloc(qv, sizeof(real));
    real *dxc = (real *)calloc(nv, sizeof(real)), *dsc = (real *)calloc(mv, sizeof(real));
    real *dzc = (real *)calloc(mv, sizeof(real)), *dyc = (real *)calloc(qv, sizeof(real));
    real *mu = (real *)calloc(B, sizeof(real)), *resid = (real *)calloc(B, sizeof(real));
    real *pri = (real *)calloc(B, sizeof(real)), *dual = (real *)calloc(B, sizeof(real));
    real *alpha = (real *)calloc(B, sizeof(real)), *rowmax = (real *)calloc(4 * (size_t)B, sizeof(real));
    int *active = (int *)calloc(B, sizeof(int)), *nnot = (int *)calloc(B, sizeof(int));
    int *floor_hit = (int *)calloc(B, sizeof(int));
    real *feas_prev = (real *)calloc(B, sizeof(real)), *alpha_prev = (real *)calloc(B, sizeof(real));
    int *ferr = (int *)calloc(B, sizeof(int));
    int have_best = 0, nNotImproved = 0, trips = 0;
    const size_t wsz = (size_t)n + m + 2 * q + 4;

    /* initial point: d = 1, factor_kkt, solve_kkt(p, 0, -h, -b)   batch.py:61-67 */
#pragma omp parallel num_threads(nthreads)
    {
        real *work = (real *)malloc(sizeof(real) * wsz);
        real *t0 = (real *)malloc(sizeof(real) * (2 * m + q + 1));
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < B; ++i) {
            real *di = d + (size_t)i * m;
            for (int k = 0; k < m; ++k) { di[k] = 1; t0[k] = 0; t0[m + k] = -h[(size_t)i * m + k]; }
            for (int k = 0; k < q; ++k) t0[2 * m + k] = -b[(size_t)i * q + k];
            factor_kkt_one(&F[i], di);
            solve_kkt_one(&F[i], G + (size_t)i * m * n, q > 0 ? A + (size_t)i * q * n : NULL, di,
                          p + (size_t)i * n, t0, t0 + m, t0 + 2 * m, x + (size_t)i * n,
                          s + (size_t)i * m, z + (size_t)i * m, y + (size_t)i * (q > 0 ? q : 1), work);
            /* make slacks and inequality duals >= 1     batch.py:76-87 */
            real *si = s + (size_t)i * m, *zi = z + (size_t)i * m;
            real mn = (real)INFINITY;
            for (int k = 0; k < m; ++k) mn = nan_min(mn, si[k]);
            if (mn < 0) for (int k = 0; k < m; ++k) si[k] -= mn - 1;
            mn = (real)INFINITY;
            for (int k = 0; k < m; ++k) mn = nan_min(mn, zi[k]);
            if (mn < 0) for (int k = 0; k < m; ++k) zi[k] -= mn - 1;
            active[i] = 1; iters[i] = 0; nnot[i] = 0;
            best_resid[i] = (real)INFINITY;
        }
        free(work); free(t0);
    }

    for (int it = 0; it < maxIter; ++it) {
        int any_active = 0;
        for (int i = 0; i < B; ++i) any_active |= active[i];
        if (!any_active) break;
        trips = it + 1;
        /* residuals, mu, d, factor_kkt          batch.py:94-113 */
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (int i = 0; i < B; ++i) {
            if (!active[i]) continue;
            const real *Qi = Q + (size_t)i * n * n, *Gi = G + (size_t)i * m * n;
            const real *Ai = q > 0 ? A + (size_t)i * q * n : NULL;
            real *xi = x + (size_t)i * n, *si = s + (size_t)i * m, *zi = z + (size_t)i * m;
            real *yi = y + (size_t)i * (q > 0 ? q : 1);
            real *rxi = rx + (size_t)i * n, *rzi = rz + (size_t)i * m, *ryi = ry + (size_t)i * (q > 0 ? q : 1);
            real nrx = 0, nrz = 0, nry = 0, sz = 0;
            for (int k = 0; k < n; ++k) {
                real acc = 0;
                for (int j = 0; j < q; ++j) acc += yi[j] * Ai[j * n + k];
                real acc2 = 0;
                for (int j = 0; j < m; ++j) acc2 += zi[j] * Gi[j * n + k];
                real acc3 = 0;
                for (int j = 0; j < n; ++j) acc3 += xi[j] * Qi[k * n + j];
                rxi[k] = acc + acc2 + acc3 + p[(size_t)i * n + k];
                nrx += rxi[k] * rxi[k];
            }
            for (int k = 0; k < m; ++k) {
                real acc = 0;
                for (int j = 0; j < n; ++j) acc += xi[j] * Gi[k * n + j];
                rzi[k] = acc + si[k] - h[(size_t)i * m + k];
                nrz += rzi[k] * rzi[k];
                sz += si[k] * zi[k];
                rs[(size_t)i * m + k] = zi[k];
            }
            for (int k = 0; k < q; ++k) {
                real acc = 0;
                for (int j = 0; j < n; ++j) acc += xi[j] * Ai[k * n + j];
                ryi[k] = acc - b[(size_t)i * q + k];
                nry += ryi[k] * ryi[k];
            }
            mu[i] = (real)fabs((double)(sz / m));
            pri[i] = (real)sqrt((double)nry) + (real)sqrt((double)nrz);
            dual[i] = (real)sqrt((double)nrx);
            resid[i] = pri[i] + dual[i] + m * mu[i];
            real *di = d + (size_t)i * m;
            for (int k = 0; k < m; ++k) di[k] = zi[k] / si[k];
            ferr[i] = factor_kkt_one(&F[i], di);
        }
        /* `except: return best` (batch.py:110-113).  Batch mode: any failure ends the solve. */
        int anyfail = 0;
        for (int i = 0; i < B; ++i) if (active[i] && ferr[i]) anyfail = 1;
        if (anyfail && !per_qp) break;
        if (trace) {
            double a0 = 0, a1 = 0, a2 = 0;
            for (int i = 0; i < B; ++i) { a0 += pri[i]; a1 += dual[i]; a2 += mu[i]; }
            trace[3 * it] = (real)(a0 / B); trace[3 * it + 1] = (real)(a1 / B); trace[3 * it + 2] = (real)(a2 / B);
        }
        /* best-iterate tracking       batch.py:118-139 */
        int improved_any = 0;
        for (int i = 0; i < B; ++i) {
            if (!active[i]) continue;
            if (per_qp && ferr[i]) { active[i] = 0; continue; }
            iters[i] = it + 1;
            int better = (it == 0) ? 1 : (resid[i] < best_resid[i]);
            if (better) {
                best_resid[i] = resid[i];
                memcpy(x_out + (size_t)i * n, x + (size_t)i * n, sizeof(real) * n);
                memcpy(z_out + (size_t)i * m, z + (size_t)i * m, sizeof(real) * m);
                memcpy(s_out + (size_t)i * m, s + (size_t)i * m, sizeof(real) * m);
                if (q > 0) memcpy(y_out + (size_t)i * q, y + (size_t)i * q, sizeof(real) * q);
                if (it > 0) improved_any = 1;
                nnot[i] = 0;
            } else if (!per_qp || stall_policy == 1 ||
                       (stall_policy == 2 && m * mu[i] < pri[i] + dual[i])) {
                nnot[i] += 1;
            } else if (stall_policy == 2) {
                nnot[i] = 0;     /* counter only runs on the feasibility floor */
            }
        }
        have_best = 1;
        if (it > 0) nNotImproved = improved_any ? 0 : nNotImproved + 1;
        /* termination      batch.py:140-143 */
        if (!per_qp) {
            real bmax = -(real)INFINITY, mumin = (real)INFINITY;
            for (int i = 0; i < B; ++i) { bmax = nan_max(bmax, best_resid[i]); mumin = nan_min(mumin, mu[i]); }
            if (nNotImproved == notImprovedLim || bmax < (real)eps || mumin > (real)1e32) break;
        } else {
            for (int i = 0; i < B; ++i) {
                if (!active[i]) continue;
                real feas = pri[i] + dual[i];
                if (stall_policy == 2 && it >= 1 && feas > (real)2 * ((real)1 - alpha_prev[i]) * feas_prev[i])
                    floor_hit[i] = 1;
                feas_prev[i] = feas;
                if ((stall_policy != 0 && nnot[i] >= notImprovedLim) || best_resid[i] < (real)eps ||
                    mu[i] > (real)1e32 || !isfinite((double)resid[i]) ||
                    (stall_policy == 2 && floor_hit[i] && m * mu[i] < (real)1e-2 * feas))
                    active[i] = 0;
            }
        }
        /* affine scaling direction      batch.py:145-151 */
#pragma omp parallel num_threads(nthreads)
        {
            real *work = (real *)malloc(sizeof(real) * wsz);
#pragma omp for schedule(dynamic, 1)
            for (int i = 0; i < B; ++i) {
                if (!active[i]) continue;
                const size_t qo = (size_t)i * (q > 0 ? q : 1);
                solve_kkt_one(&F[i], G + (size_t)i * m * n, q > 0 ? A + (size_t)i * q * n : NULL,
                              d + (size_t)i * m, rx + (size_t)i * n, rs + (size_t)i * m, rz + (size_t)i * m,
                              ry + qo, dxa + (size_t)i * n, dsa + (size_t)i * m, dza + (size_t)i * m, dya + qo, work);
                rowmax[4 * i] = get_step_rowmax(m, z + (size_t)i * m, dza + (size_t)i * m);
                rowmax[4 * i + 1] = get_step_rowmax(m, s + (size_t)i * m, dsa + (size_t)i * m);
            }
            free(work);
        }
        /* alpha_aff, sigma, corrector rhs      batch.py:160-172 */
        real gz = -(real)INFINITY, gs = -(real)INFINITY;
        if (!per_qp)
            for (int i = 0; i < B; ++i) { gz = nan_max(gz, rowmax[4 * i]); gs = nan_max(gs, rowmax[4 * i + 1]); }
#pragma omp parallel num_threads(nthreads)
        {
            real *work = (real *)malloc(sizeof(real) * wsz);
            real *zero = (real *)calloc((size_t)n + m + q + 1, sizeof(real));
#pragma omp for schedule(dynamic, 1)
            for (int i = 0; i < B; ++i) {
                if (!active[i]) continue;
                const size_t qo = (size_t)i * (q > 0 ? q : 1);
                real *si = s + (size_t)i * m, *zi = z + (size_t)i * m;
                real *dsai = dsa + (size_t)i * m, *dzai = dza + (size_t)i * m;
                real fz = py_max1(per_qp ? rowmax[4 * i] : gz), fs = py_max1(per_qp ? rowmax[4 * i + 1] : gs);
                real a = nan_min(nan_min(get_step_row(m, zi, dzai, fz), get_step_row(m, si, dsai, fs)), (real)1);
                real t3 = 0, t4 = 0;
                for (int k = 0; k < m; ++k) {
                    t3 += (si[k] + a * dsai[k]) * (zi[k] + a * dzai[k]);
                    t4 += si[k] * zi[k];
                }
                real sg = t3 / t4; sg = sg * sg * sg;
                real *rsi = rs + (size_t)i * m;
                for (int k = 0; k < m; ++k) rsi[k] = (-mu[i] * sg + dsai[k] * dzai[k]) / si[k];
                solve_kkt_one(&F[i], G + (size_t)i * m * n, q > 0 ? A + (size_t)i * q * n : NULL,
                              d + (size_t)i * m, zero, rsi, zero + n, zero + n + m, dxc + (size_t)i * n,
                              dsc + (size_t)i * m, dzc + (size_t)i * m, dyc + qo, work);
                /* dx = aff + cor   batch.py:189-192 (accumulate into the *a arrays) */
                for (int k = 0; k < n; ++k) dxa[(size_t)i * n + k] += dxc[(size_t)i * n + k];
                for (int k = 0; k < m; ++k) { dsai[k] += dsc[(size_t)i * m + k]; dzai[k] += dzc[(size_t)i * m + k]; }
                for (int k = 0; k < q; ++k) dya[qo + k] += dyc[qo + k];
                rowmax[4 * i + 2] = get_step_rowmax(m, zi, dzai);
                rowmax[4 * i + 3] = get_step_rowmax(m, si, dsai);
            }
            free(work); free(zero);
        }
        if (!per_qp) {
            gz = -(real)INFINITY; gs = -(real)INFINITY;
            for (int i = 0; i < B; ++i) { gz = nan_max(gz, rowmax[4 * i + 2]); gs = nan_max(gs, rowmax[4 * i + 3]); }
        }
        /* step   batch.py:193-203 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int i = 0; i < B; ++i) {
            if (!active[i]) continue;
            const size_t qo = (size_t)i * (q > 0 ? q : 1);
            real *si = s + (size_t)i * m, *zi = z + (size_t)i * m;
            real fz = py_max1(per_qp ? rowmax[4 * i + 2] : gz), fs = py_max1(per_qp ? rowmax[4 * i + 3] : gs);
            real a = nan_min((real)0.999 * nan_min(get_step_row(m, zi, dza + (size_t)i * m, fz),
                                                   get_step_row(m, si, dsa + (size_t)i * m, fs)), (real)1);
            alpha_prev[i] = a;
            for (int k = 0; k < n; ++k) x[(size_t)i * n + k] += a * dxa[(size_t)i * n + k];
            for (int k = 0; k < m; ++k) { si[k] += a * dsa[(size_t)i * m + k]; zi[k] += a * dza[(size_t)i * m + k]; }
            for (int k = 0; k < q; ++k) y[qo + k] += a * dya[qo + k];
        }
    }
    (void)have_best; (void)alpha;
    free(F); free(x); free(s); free(z); free(y); free(d); free(rx); free(rs); free(rz); free(ry);
    free(dxa); free(dsa); free(dza); free(dya); free(dxc); free(dsc); free(dzc); free(dyc);
    free(mu); free(resid); free(pri); free(dual); free(alpha); free(rowmax); free(active); free(nnot); free(ferr); free(floor_hit); free(feas_prev); free(alpha_prev);
    return trips;
}

/*
 * QPFunctionFn.backward (qp.py:127-182) per QP, before the `.mean(0)` of broadcast params:
 * d = clamp(lam,1e-8)/clamp(slack,1e-8); factor_kkt; solve_kkt(dl_dz,0,0,0); outer products.
 * `state` must hold pre_factor_kkt's result for this batch.  dA/db may be NULL when q == 0.
 */
void FN(qpo_backward)(int B, int n, int m, int q, const real *G, const real *A, void *state,
                      const real *zhat, const real *lam, const real *slack, const real *nu,
                      const real *dl_dz, real *dQ, real *dp, real *dG, real *dh, real *dA, real *db,
                      int nthreads)
{
    qp_factors *F = (qp_factors *)malloc(sizeof(qp_factors) * B);
    bind_batch(F, state, B, n, m, q);
#pragma omp parallel num_threads(nthreads)
    {
        real *work = (real *)malloc(sizeof(real) * ((size_t)n + m + 2 * q + 4));
        real *zero = (real *)calloc((size_t)2 * m + q + 1, sizeof(real));
        real *d = (real *)malloc(sizeof(real) * m), *dx = (real *)malloc(sizeof(real) * n);
        real *ds = (real *)malloc(sizeof(real) * m), *dz = (real *)malloc(sizeof(real) * m);
        real *dy = (real *)malloc(sizeof(real) * (q > 0 ? q : 1));
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < B; ++i) {
            const real *li = lam + (size_t)i * m, *sl = slack + (size_t)i * m, *zh = zhat + (size_t)i * n;
            for (int k = 0; k < m; ++k) {
                real a = li[k] < (real)1e-8 ? (real)1e-8 : li[k];
                real c = sl[k] < (real)1e-8 ? (real)1e-8 : sl[k];
                d[k] = a / c;                                            /* qp.py:148 */
            }
            factor_kkt_one(&F[i], d);                                      /* qp.py:150 */
            solve_kkt_one(&F[i], G + (size_t)i * m * n, q > 0 ? A + (size_t)i * q * n : NULL, d,
                          dl_dz + (size_t)i * n, zero, zero + m, zero + 2 * m, dx, ds, dz, dy, work);
            for (int k = 0; k < n; ++k) dp[(size_t)i * n + k] = dx[k];    /* qp.py:157 */
            for (int r = 0; r < m; ++r) {                                  /* qp.py:158,161 */
                for (int k = 0; k < n; ++k) dG[((size_t)i * m + r) * n + k] = dz[r] * zh[k] + li[r] * dx[k];
                dh[(size_t)i * m + r] = -dz[r];
            }
            for (int r = 0; r < q; ++r) {                                  /* qp.py:165-166 */
                for (int k = 0; k < n; ++k)
                    dA[((size_t)i * q + r) * n + k] = dy[r] * zh[k] + nu[(size_t)i * q + r] * dx[k];
                db[(size_t)i * q + r] = -dy[r];
            }
            for (int r = 0; r < n; ++r)                                    /* qp.py:173 */
                for (int k = 0; k < n; ++k)
                    dQ[((size_t)i * n + r) * n + k] = (real)0.5 * (dx[r] * zh[k] + zh[r] * dx[k]);
        }
        free(work); free(zero); free(d); free(dx); free(ds); free(dz); free(dy);
    }
    free(F);
}
