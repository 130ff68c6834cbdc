// qpx_kernels.h -- device bodies of the batched dense QP solver (one QP per workgroup).
//
// What the reference does with ~230 generic batched tensor ops per IPM iteration
// (qpth/solvers/pdipm/batch.py:47-207) is done here by three kernels:
//
//   prefactor_body ..... pre_factor_kkt (batch.py:375-429) + the constant part of the start
//                        point: Cholesky of Q, Zp = P L^-1 G^T, R = Zp^T Zp, equality block
//   ipm_body ........... factor_kkt (batch.py:435-470), solve_kkt (batch.py:349-372) and the
//                        whole PDIPM loop (batch.py:61-207), in the m-dimensional "condensed"
//                        space: the iterate x is never formed inside the loop because
//                        x = x0 - M^T z' holds for every iterate
//   kkt_body ........... one factor_kkt + solve_kkt for arbitrary right-hand sides (the unit
//                        the reference tests in test.py:222-234) and, with kBackward, the
//                        gradient epilogue of QPFunctionFn.backward (qp.py:127-182)
//
// All three are templates over the scalar type, the number of 64-lane "slots" a vector of
// length max(n,m,q) needs (NS) and whether the matrices are staged in LDS (kLds) or worked on
// in place in the HBM factor blob (sizes that exceed 160 KiB of LDS).
#pragma once
#include "qpx_layout.h"
#include "qpx_platform.h"

// Phase timers of the profiling build (-DQPX_PROFILE, scripts/prof_phases.py): thread 0 of every
// workgroup accumulates shader-clock cycles per phase and writes 8 numbers per QP into `trace`.
#ifdef QPX_PROFILE
#define QPX_PROF_INIT long long qpx_prof_t = clock64(); long long qpx_prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define QPX_PROF(i) { const long long qpx_prof_n = clock64(); qpx_prof_acc[i] += qpx_prof_n - qpx_prof_t; qpx_prof_t = qpx_prof_n; }
#define QPX_PROF_DUMP(ptr, T) if (b.tid == 0 && (ptr)) { for (int qpx_i = 0; qpx_i < 8; ++qpx_i) (ptr)[qpx_i] = (T)qpx_prof_acc[qpx_i]; }
#else
#define QPX_PROF_INIT
#define QPX_PROF(i)
#define QPX_PROF_DUMP(ptr, T)
#endif

namespace qpx {

template <class T> struct Lim;
template <> struct Lim<float> {
    static QPX_DEV float inf() { return __builtin_huge_valf(); }
    static QPX_DEV float tiny() { return 1.17549435e-38f; }
};
template <> struct Lim<double> {
    static QPX_DEV double inf() { return __builtin_huge_val(); }
    static QPX_DEV double tiny() { return 2.2250738585072014e-308; }
};

// ------------------------------------------------------------------------------------------
// wave-level helpers (every lane of the calling wave takes part)

// Reductions over the 64 lanes; every lane gets the result.  Inside the four 16-lane rows with DPP
// moves, across the rows with four v_readlane broadcasts: ~25 VALU issues and no LDS traffic (the
// ds_bpermute butterfly these replace cost ~600 ticks per reduction on MI355X).
template <class T> QPX_DEV T wave_sum(const Block& b, T v)
{
    v += b.template xor16<1>(v);
    v += b.template xor16<2>(v);
    v += b.template xor16<7>(v);
    v += b.template xor16<15>(v);
    return (b.bcast(v, 0) + b.bcast(v, 16)) + (b.bcast(v, 32) + b.bcast(v, 48));
}
template <class T> QPX_DEV T min2_(T a, T c) { return (c < a) ? c : a; }
template <class T> QPX_DEV T max2_(T a, T c) { return (c > a) ? c : a; }
template <class T> QPX_DEV T wave_min(const Block& b, T v)
{
    v = min2_(v, b.template xor16<1>(v));
    v = min2_(v, b.template xor16<2>(v));
    v = min2_(v, b.template xor16<7>(v));
    v = min2_(v, b.template xor16<15>(v));
    return min2_(min2_(b.bcast(v, 0), b.bcast(v, 16)), min2_(b.bcast(v, 32), b.bcast(v, 48)));
}
template <class T> QPX_DEV T wave_max(const Block& b, T v)
{
    v = max2_(v, b.template xor16<1>(v));
    v = max2_(v, b.template xor16<2>(v));
    v = max2_(v, b.template xor16<7>(v));
    v = max2_(v, b.template xor16<15>(v));
    return max2_(max2_(b.bcast(v, 0), b.bcast(v, 16)), max2_(b.bcast(v, 32), b.bcast(v, 48)));
}

// A vector of length n is held by ONE wave as NS registers per lane: element i lives in
// slot i/64 of lane i%64.
template <int NS, class T> QPX_DEV void ld_slots(const Block& b, T (&x)[NS], const T* src, int n, T fill)
{
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = s * kWave + b.lane();
        x[s] = (i < n) ? src[i] : fill;
    }
}
template <int NS, class T> QPX_DEV void st_slots(const Block& b, T* dst, const T (&x)[NS], int n)
{
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = s * kWave + b.lane();
        if (i < n) dst[i] = x[s];
    }
}

// Forward substitution L y = x (in place) by one wave.  L is packed lower row-major
// (L_ik = P[tri(i)+k]), dinv[k] = 1/L_kk.  Column oriented: once y_k is final it is
// broadcast with v_readlane (no LDS round trip) and every lane updates the rows it owns.
// The L column entries of UNR consecutive steps are fetched up front so the LDS/L2 latency
// is paid once per UNR steps and the dependent chain per step is readlane + mul + fma.
template <int NS, class T> QPX_DEV void trsv_fwd(const Block& b, const T* P, const T* dinv, int n, T (&x)[NS])
{
    constexpr int UNR = 4;
    const int lane = b.lane();
#pragma unroll
    for (int sk = 0; sk < NS; ++sk) {
        const int kbase = sk * kWave;
        if (kbase < n) {
            const int kend = (n - kbase < kWave) ? (n - kbase) : kWave;
            for (int lk0 = 0; lk0 < kend; lk0 += UNR) {
                T l[UNR][NS];
                T di[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int k = kbase + lk0 + u;
                    di[u] = (lk0 + u < kend) ? dinv[k] : T(0);
#pragma unroll
                    for (int s2 = sk; s2 < NS; ++s2) {
                        const int i = s2 * kWave + lane;
                        l[u][s2] = (lk0 + u < kend && i > k && i < n) ? P[tri(i) + k] : T(0);
                    }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int lk = lk0 + u;
                    if (lk < kend) {
                        const T yk = b.bcast(x[sk], lk) * di[u];
                        if (lane == lk) x[sk] = yk;
#pragma unroll
                        for (int s2 = sk; s2 < NS; ++s2) x[s2] = fma_(-l[u][s2], yk, x[s2]);
                    }
                }
            }
        }
    }
}

// Backward substitution L^T y = x (in place) by one wave; reads ROW k of L per step
// (contiguous in the packed layout).
template <int NS, class T> QPX_DEV void trsv_bwd(const Block& b, const T* P, const T* dinv, int n, T (&x)[NS])
{
    constexpr int UNR = 4;
    const int lane = b.lane();
#pragma unroll
    for (int sk = NS - 1; sk >= 0; --sk) {
        const int kbase = sk * kWave;
        if (kbase < n) {
            const int kend = (n - kbase < kWave) ? (n - kbase) : kWave;
            for (int lk0 = kend - 1; lk0 >= 0; lk0 -= UNR) {
                T l[UNR][NS];
                T di[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int lk = lk0 - u;
                    const int k = kbase + lk;
                    di[u] = (lk >= 0) ? dinv[k] : T(0);
#pragma unroll
                    for (int s2 = 0; s2 <= sk; ++s2) {
                        const int i = s2 * kWave + lane;
                        l[u][s2] = (lk >= 0 && i < k) ? P[tri(k) + i] : T(0);
                    }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int lk = lk0 - u;
                    if (lk >= 0) {
                        const T yk = b.bcast(x[sk], lk) * di[u];
                        if (lane == lk) x[sk] = yk;
#pragma unroll
                        for (int s2 = 0; s2 <= sk; ++s2) x[s2] = fma_(-l[u][s2], yk, x[s2]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// workgroup-level helpers

template <class T> QPX_DEV void block_copy(const Block& b, T* dst, const T* src, size_t count)
{
    for (size_t e = b.tid; e < count; e += b.nt) dst[e] = src[e];
}

// In-place Cholesky of the packed lower-triangular n x n matrix P (row-major packed) by the
// whole workgroup.  Right-looking with ONE barrier per column: the trailing update uses the
// un-scaled pivot column, T_ij -= T_ik T_jk / T_kk, so no thread has to publish a scaled
// column first; a final pass scales column k by 1/sqrt(d_k).  dinv[k] = 1/L_kk on return.
// Returns false (uniformly) on a non-positive or non-finite pivot.
// Replaces lu_hack (batch.py:8-20; un-pivoted on the reference's GPU branch).
template <class T> QPX_DEV bool chol_packed(const Block& b, T* P, int n, T* dinv)
{
    constexpr int TJ = 16;
    const int TI = b.nt / TJ;
    const int ti = b.tid / TJ, tj = b.tid % TJ;
    for (int k = 0; k < n; ++k) {
        b.sync();
        const T dk = P[tri(k) + k];
        if (!(dk > T(0)) || !finite_(dk)) return false;
        const T rk = T(1) / dk;
        for (int i = k + 1 + ti; i < n; i += TI) {
            T* Pi = P + tri(i);
            const T lik = Pi[k] * rk;
            // four independent read-modify-writes in flight per pass (the LDS latency, not the
            // FMA rate, bounds this loop)
            for (int j0 = k + 1 + tj; j0 <= i; j0 += 4 * TJ) {
                T pij[4], pjk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u * TJ;
                    if (j <= i) {
                        pij[u] = Pi[j];
                        pjk[u] = P[tri(j) + k];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u * TJ;
                    if (j <= i) Pi[j] = fma_(-lik, pjk[u], pij[u]);
                }
            }
        }
    }
    b.sync();
    for (int k = b.tid; k < n; k += b.nt) dinv[k] = T(1) / sqrt_(P[tri(k) + k]);
    b.sync();
    for (int i = ti; i < n; i += TI) {
        T* Pi = P + tri(i);
        for (int j = tj; j <= i; j += TJ) Pi[j] *= dinv[j];
    }
    b.sync();
    return true;
}

// y = S x for the symmetric m x m matrix S given as packed lower; x, y are m-vectors in LDS.
template <class T> QPX_DEV void block_symv_packed(const Block& b, const T* P, int m, const T* x, T* y)
{
    for (int i = b.tid; i < m; i += b.nt) {
        const T* Pi = P + tri(i);
        T acc = 0;
        for (int j = 0; j <= i; ++j) acc = fma_(Pi[j], x[j], acc);
        size_t t = tri(i + 1) + i;
        for (int j = i + 1; j < m; ++j) {
            acc = fma_(P[t], x[j], acc);
            t += j + 1;
        }
        y[i] = acc;
    }
}

// X <- L^-1 X for the n x ncols matrix X (row-major, leading dim ldx): every column is an
// independent forward substitution.  TPC lanes share one column (partial dot products summed
// with shuffles); the lanes of a group sit in one wave, so no workgroup barrier is needed.
template <class T>
QPX_DEV void block_trsm_lower(const Block& b, const T* P, const T* dinv, int n, T* X, int ldx, int ncols)
{
    int tpc = 1;
    while (tpc < 16 && ncols * (tpc * 2) <= b.nt) tpc *= 2;
    const int ngroups = b.nt / tpc;
    const int grp = b.tid / tpc, part = b.tid % tpc;
    // every lane of a wave must execute the same number of column passes (shuffles inside)
    const int passes = (ncols + ngroups - 1) / ngroups;
    for (int ps = 0; ps < passes; ++ps) {
        const int j = ps * ngroups + grp;
        const bool act = j < ncols;
        for (int i = 0; i < n; ++i) {
            const T* Pi = P + tri(i);
            T acc = 0;
            if (act)
                for (int k = part; k < i; k += tpc) acc = fma_(Pi[k], X[(size_t)k * ldx + j], acc);
            for (int o = 1; o < tpc; o <<= 1) acc += b.shfl_xor(acc, o);
            if (act && part == 0) X[(size_t)i * ldx + j] = (X[(size_t)i * ldx + j] - acc) * dinv[i];
            b.wave_sync();
        }
    }
}

// ------------------------------------------------------------------------------------------
// Arrays at the C boundary of a FLOAT64 kernel may be float32 (QPX_F32_WIDE in include/qpx.h: the caller keeps float32
// tensors, factors and arithmetic are float64; `io32` in the argument blocks below).  In<T> reads, put_ writes such an
// array through its declared pointer type; the branch is wave-uniform and sits outside every inner loop.  The float32
// kernels never set io32.
template <class T> struct In {
    const T* p;
    int f32;
    QPX_DEV In(const T* base, size_t off, int io32) : p(nullptr), f32(sizeof(T) == 8 ? io32 : 0)
    {
        if (base) p = f32 ? reinterpret_cast<const T*>(reinterpret_cast<const float*>(base) + off) : base + off;
    }
    QPX_DEV explicit operator bool() const { return p != nullptr; }
    QPX_DEV T operator[](size_t i) const { return f32 ? (T) reinterpret_cast<const float*>(p)[i] : p[i]; }
};
template <class T> QPX_DEV void put_(T* base, int io32, size_t i, T v)
{
    if (sizeof(T) == 8 && io32) reinterpret_cast<float*>(base)[i] = (float)v;
    else base[i] = v;
}

// ------------------------------------------------------------------------------------------
// kernel argument blocks (plain data, passed by value)

template <class T> struct PrefactorArgs {
    int B, n, m, q;
    const T *Q, *G, *A;                   // batch-major, row-major; stride 0 = shared by the batch
    long long sQ, sG, sA;                 // batch strides in elements
    T* fac;
    size_t fac_stride;
    int* status;
    int images;                           // blob family / register images of R (qpx_layout.h: fac_layout)
    int io32 = 0;                         // T = double only: Q, G, A are float32 arrays (QPX_F32_WIDE)
    // matrix-core form (qpx_prefac.h) only: which tiles of K (pf_k) and of R (pf_r) each of the four waves computes, bit
    // t = tile (i, j), t = i (i + 1) / 2 + j -- a greedy balance the host works out once per launch (prefac_deal)
    unsigned pf_k[4] = {0, 0, 0, 0}, pf_r[4] = {0, 0, 0, 0};
};

template <class T> struct IpmArgs {
    int B, n, m, q;
    const T *p, *h, *b;                   // (B,n) (B,m) (B,q); stride 0 = shared by the batch
    long long sp, sh, sb;
    T* fac;
    size_t fac_stride;
    T eps;
    int maxIter, notImprovedLim, stall_policy;
    T *zhat, *nu, *lam, *slack;           // (B,n) (B,q) (B,m) (B,m)
    int *iters, *status;
    T* best_resid;
    T* trace;                             // optional [maxIter][B][3]: pri_resid, dual_resid, mu
    int images;                           // blob family the factors were written in (fac_layout)
    int io32 = 0;                         // T = double only: every array but `fac` is float32 (QPX_F32_WIDE)
};

template <class T> struct KktArgs {
    int B, n, m, q;
    T* fac;
    size_t fac_stride;
    // general solve_kkt (batch.py:349-372): d (B,m), rx (B,n), rs, rz (B,m), ry (B,q); NULL = zeros
    const T *d, *rx, *rs, *rz, *ry;
    T *dx, *ds, *dz, *dy;
    // backward (qp.py:127-182): d is built from lam/slack, rx = dl_dz
    const T *zhat, *lam, *slack, *nu, *dl_dz;
    T *dQ, *dp, *dG, *dh, *dA, *db;       // backward: any of them may be NULL (gradient not wanted)
    int* status;
    int images;                           // blob family the factors were written in (fac_layout)
    // iterative refinement on the residual of the original KKT system (batch.py:244-270): steps, and the caller's
    // Q (B,n,n), G (B,m,n), A (B,q,n) with their batch strides; refine = 0 or Q = NULL: none
    int refine;
    const T *Q, *G, *A;
    long long sQ, sG, sA;
    int io32 = 0;                         // T = double only: every array but `fac` is float32 (QPX_F32_WIDE; refine = 0)
};

// The finishing stage (qpx_polish, include/qpx.h): iterations of the reference's loop in the ORIGINAL variables
// (batch.py:92-198) on the residuals of the caller's data, started from a given iterate, best iterate kept
template <class T> struct PolishArgs {
    int B, n, m, q;
    T* fac;
    size_t fac_stride;
    int images;
    const T *Q, *G, *A;                   // the caller's problem data (B,n,n) (B,m,n) (B,q,n); stride 0 = shared
    long long sQ, sG, sA;
    const T *p, *h, *b;                   // (B,n) (B,m) (B,q)
    long long sp, sh, sb;
    T *zhat, *nu, *lam, *slack;           // in: the iterate to start from; out: the best iterate met
    int steps, refine;
    T* best_resid;                        // out, may be NULL: the reference's residual of the returned iterate (batch.py:103-107)
    int* status;
};

constexpr size_t kMaxLdsBytes = 160 * 1024;   // gfx950: 160 KiB of LDS per workgroup

// LDS requirements (elements of T); the host uses the same formulas to size the launch.
QPX_LAYOUT_HD size_t max2(size_t a, size_t b) { return a > b ? a : b; }
QPX_LAYOUT_HD size_t lds_elems_ipm(int n, int m, int q, bool lds_mats)
{
    const size_t v = align4(max2(max2((size_t)n, (size_t)m), (size_t)q));
    return (lds_mats ? align4(max2(tri(m), tri(n))) : 0) + 8 * v + 4;
}
QPX_LAYOUT_HD size_t lds_elems_kkt(int n, int m, int q, bool lds_mats)
{
    const size_t v = align4(max2(max2((size_t)n, (size_t)m), (size_t)q));
    return (lds_mats ? align4(max2(tri(m), tri(n))) : 0) + 8 * v + 4;
}
QPX_LAYOUT_HD size_t lds_elems_prefactor(int n, int m, int q, bool lds_mats)
{
    const size_t v = align4(max2(max2((size_t)n, (size_t)m), (size_t)q));
    size_t e = 3 * v + 4;
    if (lds_mats)
        e += align4(tri(n)) + align4((size_t)n * align4(m)) + align4((size_t)n * q) +
             align4((size_t)q * align4(m)) + align4(tri(q));
    return e;
}

// ------------------------------------------------------------------------------------------
// prefactor: everything that depends only on (Q, G, A).   batch.py:375-429.
template <class T, int NS, bool kLds>
QPX_DEV void prefactor_body(const Block& b, const PrefactorArgs<T>& a, int qp, T* lds)
{
    const int n = a.n, m = a.m, q = a.q;
    const FacLayout lay = fac_layout(n, m, q, 0);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const T* Qg = a.Q + (size_t)qp * a.sQ;
    const T* Gg = a.G + (size_t)qp * a.sG;
    const T* Ag = q > 0 ? a.A + (size_t)qp * a.sA : nullptr;
    const size_t v = align4(max2(max2((size_t)n, (size_t)m), (size_t)q));

    // carve LDS
    T* vu = lds;            // scratch               (n)
    T* dq = vu + v;         // dinv of L             (n)
    T* d11 = dq + v;        // dinv of L11           (q)
    T* ctrl = d11 + v;
    T* mats = ctrl + 4;
    T *Lq, *Z, *Yh, *V, *L11;
    int ldz;
    if (kLds) {
        ldz = (int)align4(m);
        Lq = mats;
        Z = Lq + align4(tri(n));
        Yh = Z + align4((size_t)n * ldz);
        V = Yh + align4((size_t)n * q);
        L11 = V + align4((size_t)q * ldz);
    } else {
        ldz = m;
        Lq = F + lay.L; Z = F + lay.Zp; Yh = F + lay.Yh; V = F + lay.V; L11 = F + lay.L11;
    }

    QPX_PROF_INIT
    // A. symmetrised lower triangle of Q -> packed
    for (int idx = b.tid; idx < n * n; idx += b.nt) {
        const int i = idx / n, j = idx - i * n;
        if (j <= i) Lq[tri(i) + j] = T(0.5) * (Qg[(size_t)i * n + j] + Qg[(size_t)j * n + i]);
    }
    // B. Cholesky of Q
    const bool okQ = chol_packed(b, Lq, n, dq);
    if (!okQ) {
        // leave a well-defined (zero) blob behind and flag the QP
        for (size_t e = b.tid; e < lay.total; e += b.nt) F[e] = T(0);
        if (b.tid == 0) a.status[qp] = QPX_ST_Q_NOT_SPD;
        return;
    }
    QPX_PROF(0)
    // C. Z = G^T  (n x m), padded columns zero
    for (int idx = b.tid; idx < n * ldz; idx += b.nt) {
        const int i = idx / ldz, j = idx - i * ldz;
        Z[(size_t)i * ldz + j] = (j < m) ? Gg[(size_t)j * n + i] : T(0);
    }
    if (q > 0)
        for (int idx = b.tid; idx < n * q; idx += b.nt) {
            const int i = idx / q, c = idx - i * q;
            Yh[(size_t)i * q + c] = Ag[(size_t)c * n + i];
        }
    b.sync();
    // || G^T 1 ||
    for (int i = b.tid; i < n; i += b.nt) {
        T acc = 0;
        for (int j = 0; j < m; ++j) acc += Z[(size_t)i * ldz + j];
        vu[i] = acc;
    }
    b.sync();
    if (b.wave() == 0) {
        T acc = 0;
        for (int i = b.lane(); i < n; i += kWave) acc = fma_(vu[i], vu[i], acc);
        acc = wave_sum(b, acc);
        if (b.lane() == 0) F[lay.scal] = sqrt_(acc);
    }
    QPX_PROF(1)
    // D. Z <- L^-1 G^T ; Y <- L^-1 A^T
    block_trsm_lower(b, Lq, dq, n, Z, ldz, m);
    if (q > 0) block_trsm_lower(b, Lq, dq, n, Yh, q, q);
    b.sync();
    QPX_PROF(2)
    int okA = 1;
    if (q > 0) {
        // E. S11 = Y^T Y -> L11 ; Yh = Y L11^-T ; V = Yh^T Z ; Zp = Z - Yh V
        for (int idx = b.tid; idx < q * q; idx += b.nt) {
            const int r = idx / q, c = idx - r * q;
            if (c <= r) {
                T acc = 0;
                for (int k = 0; k < n; ++k) acc = fma_(Yh[(size_t)k * q + r], Yh[(size_t)k * q + c], acc);
                L11[tri(r) + c] = acc;
            }
        }
        okA = chol_packed(b, L11, q, d11) ? 1 : 0;
        if (!okA) {
            for (size_t e = b.tid; e < lay.total; e += b.nt) F[e] = T(0);
            if (b.tid == 0) a.status[qp] = QPX_ST_A_RANK;
            return;
        }
        for (int r = b.tid; r < n; r += b.nt) {
            T* yr = Yh + (size_t)r * q;
            for (int c = 0; c < q; ++c) {
                T acc = yr[c];
                const T* Lc = L11 + tri(c);
                for (int k = 0; k < c; ++k) acc = fma_(-yr[k], Lc[k], acc);
                yr[c] = acc * d11[c];
            }
        }
        b.sync();
        for (int idx = b.tid; idx < q * ldz; idx += b.nt) {
            const int r = idx / ldz, j = idx - r * ldz;
            T acc = 0;
            for (int k = 0; k < n; ++k) acc = fma_(Yh[(size_t)k * q + r], Z[(size_t)k * ldz + j], acc);
            V[(size_t)r * ldz + j] = acc;
        }
        b.sync();
        for (int idx = b.tid; idx < n * ldz; idx += b.nt) {
            const int k = idx / ldz, j = idx - k * ldz;
            T acc = Z[(size_t)k * ldz + j];
            for (int r = 0; r < q; ++r) acc = fma_(-Yh[(size_t)k * q + r], V[(size_t)r * ldz + j], acc);
            Z[(size_t)k * ldz + j] = acc;
        }
        b.sync();
    }
    QPX_PROF(3)
    // F. R = Zp^T Zp, packed lower, straight to HBM; 4x4 register tiles
    {
        T* Rg = F + lay.R;
        const int mt = (m + 3) / 4;
        for (int t = b.tid; t < mt * mt; t += b.nt) {
            const int ti = t / mt, tj = t - ti * mt;
            if (tj > ti) continue;
            T acc[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = 0;
            for (int k = 0; k < n; ++k) {
                const T* zk = Z + (size_t)k * ldz;
                T ar[4], bc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ar[r] = (4 * ti + r < m) ? zk[4 * ti + r] : T(0);
                    bc[r] = (4 * tj + r < m) ? zk[4 * tj + r] : T(0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = fma_(ar[r], bc[c], acc[r][c]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int i = 4 * ti + r, j = 4 * tj + c;
                    if (i < m && j <= i) {
                        Rg[tri(i) + j] = acc[r][c];
                    }
                }
        }
    }
    b.sync();
    QPX_PROF(4)
    // G. r1 = R 1 = Zp^T (Zp 1)
    for (int k = b.tid; k < n; k += b.nt) {
        T acc = 0;
        for (int j = 0; j < m; ++j) acc += Z[(size_t)k * ldz + j];
        vu[k] = acc;
    }
    b.sync();
    for (int j = b.tid; j < m; j += b.nt) {
        T acc1 = 0;
        for (int k = 0; k < n; ++k) acc1 = fma_(Z[(size_t)k * ldz + j], vu[k], acc1);
        F[lay.r1 + j] = acc1;
    }
    // H. spill the staged matrices to the blob
    for (int k = b.tid; k < n; k += b.nt) F[lay.dinvL + k] = dq[k];
    for (int r = b.tid; r < q; r += b.nt) F[lay.dinv11 + r] = d11[r];
    if (kLds) {
        block_copy(b, F + lay.L, Lq, tri(n));
        for (int idx = b.tid; idx < n * m; idx += b.nt) {
            const int k = idx / m, j = idx - k * m;
            F[lay.Zp + idx] = Z[(size_t)k * ldz + j];
        }
        if (q > 0) {
            block_copy(b, F + lay.Yh, Yh, (size_t)n * q);
            for (int idx = b.tid; idx < q * m; idx += b.nt) {
                const int r = idx / m, j = idx - r * m;
                F[lay.V + idx] = V[(size_t)r * ldz + j];
            }
            block_copy(b, F + lay.L11, L11, tri(q));
        }
    }
    if (b.tid == 0) a.status[qp] = 0;
    b.sync();
    QPX_PROF(5)
    QPX_PROF_DUMP(F + lay.prof, T)
}

// ------------------------------------------------------------------------------------------
// step length to the boundary for one QP (get_step, batch.py:210-213, as it behaves for a
// batch of one): min over blocking entries of -v/dv, or 1 when nothing blocks.
template <int NS, class T>
QPX_DEV T step_to_boundary(const Block& b, const T (&v)[NS], const T (&dv)[NS], int m)
{
    T r = Lim<T>::inf();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = s * kWave + b.lane();
        if (i < m && dv[s] < T(0)) {
            const T t = -v[s] / dv[s];
            r = (t < r) ? t : r;
        }
    }
    r = wave_min(b, r);
    return (r == Lim<T>::inf()) ? T(1) : r;
}

// The same step length from the reciprocals rv = 1/v the caller already holds: the minimum of -v/dv
// is the reciprocal of the maximum of -dv/v, so the per-entry divisions become multiplications and
// one division remains (f64 division is a ~15-instruction sequence on gfx950).
template <int NS, class T>
QPX_DEV T step_to_boundary_rcp(const Block& b, const T (&rv)[NS], const T (&dv)[NS], int m)
{
    T t = T(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = s * kWave + b.lane();
        if (i < m && dv[s] < T(0)) t = max2_(t, -dv[s] * rv[s]);
    }
    t = wave_max(b, t);
    return (t > T(0)) ? T(1) / t : T(1);
}

// Solve T dz = -rhs with the factored T = L_T L_T^T (wave 0).
template <int NS, class T>
QPX_DEV void solve_neg(const Block& b, const T* Lt, const T* dinv, int m, T (&x)[NS])
{
    trsv_fwd<NS>(b, Lt, dinv, m, x);
    trsv_bwd<NS>(b, Lt, dinv, m, x);
#pragma unroll
    for (int s = 0; s < NS; ++s) x[s] = -x[s];
}

// ------------------------------------------------------------------------------------------
// The PDIPM loop (batch.py:47-207) for one QP, in the condensed space.
//
//   state: z (lam), s (slacks), tau.  z' = z - tau*sigz*1 is the dual the implicit primal
//   iterate corresponds to (x = x0 - M^T z', nu = nu0 - W^T z'); sigz/sigs are the start-point
//   shifts of batch.py:76-87.  Residuals: rx = tau*sigz*G^T 1, ry = 0, rz = s - c - R z'.
//   Newton systems: (R + diag(s/z)) dz = -(rhs), rhs_aff = c + R z, rhs_cor = rs_cor * s/z.
template <class T, int NS, bool kLds>
QPX_DEV void ipm_body(const Block& b, const IpmArgs<T>& a, int qp, T* lds)
{
    const int n = a.n, m = a.m, q = a.q;
    const FacLayout lay = fac_layout(n, m, q, 0);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const T* Rg = F + lay.R;
    const size_t v = align4(max2(max2((size_t)n, (size_t)m), (size_t)q));
    T* dinv = lds;
    T* vA = dinv + v;   // z' (m)
    T* vB = vA + v;     // R z' (m) ; later w (n)
    T* vC = vB + v;     // c = h - G x0 (m)
    T* vW = vC + v;     // w0: x0 = L^-T w0 is the minimiser without inequalities (n)
    T* vT = vW + v;     // t = L^-1 p (n)
    T* vQ1 = vT + v;    // Yh^T t, then ycoef = beta + Yh^T t (q)
    T* vQ2 = vQ1 + v;   // beta = L11^-1 b (q)
    int* ctrl = reinterpret_cast<int*>(vQ2 + v);
    T* Tm = kLds ? (vQ2 + v + 4) : (F + lay.T);

    const int lane = b.lane();
    const bool w0 = b.wave() == 0;
    const T mT = (T)m;
    const T* pg = a.p + (size_t)qp * a.sp;
    const T* hg = a.h + (size_t)qp * a.sh;
    const T* bg = q > 0 ? a.b + (size_t)qp * a.sb : nullptr;

    // a QP whose pre-factorisation failed (Q not SPD / A rank deficient) is not iterated:
    // NaN outputs, status left for the host to raise on (qp.py:85, batch.py:381-386)
    if (a.status[qp] & (QPX_ST_Q_NOT_SPD | QPX_ST_A_RANK)) {
        const T nanv = Lim<T>::inf() - Lim<T>::inf();
        for (int i = b.tid; i < n; i += b.nt) a.zhat[(size_t)qp * n + i] = nanv;
        for (int i = b.tid; i < m; i += b.nt) {
            a.lam[(size_t)qp * m + i] = nanv;
            a.slack[(size_t)qp * m + i] = nanv;
        }
        for (int i = b.tid; i < q; i += b.nt) a.nu[(size_t)qp * q + i] = nanv;
        if (b.tid == 0) {
            a.iters[qp] = 0;
            a.best_resid[qp] = Lim<T>::inf();
        }
        return;
    }

    T z[NS], s[NS], c[NS], r1[NS], bz[NS], bs[NS], rzp[NS];
    T tau = 1, btau = 1, sigz = 0, sigs = 0, bres = Lim<T>::inf(), g1n = 0;
    T feas_prev = 0, alpha_prev = 0;
    int nnot = 0, floor_hit = 0, st = 0, iters = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) { z[k] = s[k] = bz[k] = bs[k] = T(1); c[k] = r1[k] = rzp[k] = T(0); }

    QPX_PROF_INIT
    // ---- constants that depend on p, h, b (they enter the reference through the start-point
    //      solve, batch.py:64-67):  t = L^-1 p, beta = L11^-1 b,
    //      w0 = -(t - Yh Yh^T t) + Yh beta,  c = h + Zp^T t - V^T beta,  ycoef = beta + Yh^T t
    {
        const T* Zp = F + lay.Zp;
        const T* Yh = F + lay.Yh;
        const T* V = F + lay.V;
        const T* Lp0 = F + lay.L;
        const T* dLp0 = F + lay.dinvL;
        if (kLds) {
            block_copy(b, Tm, F + lay.L, tri(n));
            block_copy(b, dinv, F + lay.dinvL, (size_t)n);
            Lp0 = Tm;
            dLp0 = dinv;
        }
        b.sync();
        if (w0) {
            T x[NS];
            ld_slots<NS>(b, x, pg, n, T(0));
            trsv_fwd<NS>(b, Lp0, dLp0, n, x);
            st_slots<NS>(b, vT, x, n);
            if (q > 0) {
                T r[NS];
                ld_slots<NS>(b, r, bg, q, T(0));
                trsv_fwd<NS>(b, F + lay.L11, F + lay.dinv11, q, r);
                st_slots<NS>(b, vQ2, r, q);
            }
        }
        b.sync();
        for (int r = b.tid; r < q; r += b.nt) {
            T acc = 0;
            for (int k = 0; k < n; ++k) acc = fma_(Yh[(size_t)k * q + r], vT[k], acc);
            vQ1[r] = acc;
        }
        b.sync();
        for (int k = b.tid; k < n; k += b.nt) {
            T w = -vT[k];
            for (int r = 0; r < q; ++r) w = fma_(Yh[(size_t)k * q + r], vQ1[r] + vQ2[r], w);
            vW[k] = w;
        }
        for (int j = b.tid; j < m; j += b.nt) {
            T acc = hg[j];
            for (int k = 0; k < n; ++k) acc = fma_(Zp[(size_t)k * m + j], vT[k], acc);
            for (int r = 0; r < q; ++r) acc = fma_(-V[(size_t)r * m + j], vQ2[r], acc);
            vC[j] = acc;
        }
        b.sync();
        for (int r = b.tid; r < q; r += b.nt) vQ1[r] += vQ2[r];
    }

    QPX_PROF(0)
    // ---- start point: d = 1 (batch.py:61-67): z_i = -(R + I)^-1 c, s_i = -z_i, then shifts
    block_copy(b, Tm, Rg, tri(m));
    b.sync();
    if (w0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) Tm[tri(i) + i] += T(1);
        }
    }
    bool ok = chol_packed(b, Tm, m, dinv);   // first statement inside is a barrier
    if (w0) {
        ld_slots<NS>(b, c, vC, m, T(0));
        ld_slots<NS>(b, r1, F + lay.r1, m, T(0));
        g1n = F[lay.scal];
        if (ok) {
            T x[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) x[k] = c[k];
            solve_neg<NS>(b, Tm, dinv, m, x);   // x = z_i
            T mnz = Lim<T>::inf(), mns = Lim<T>::inf();
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                if (i < m) {
                    mnz = (x[k] < mnz) ? x[k] : mnz;
                    mns = (-x[k] < mns) ? -x[k] : mns;
                }
            }
            mnz = wave_min(b, mnz);
            mns = wave_min(b, mns);
            sigz = (mnz < T(0)) ? (T(1) - mnz) : T(0);     // batch.py:82-87
            sigs = (mns < T(0)) ? (T(1) - mns) : T(0);     // batch.py:76-80
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                if (i < m) {
                    z[k] = x[k] + sigz;
                    s[k] = -x[k] + sigs;
                    vA[i] = x[k];                            // z' = z - tau*sigz (tau = 1)
                }
                bz[k] = z[k];
                bs[k] = s[k];
            }
        } else {
            st |= QPX_ST_KKT_BREAKDOWN;
        }
        if (lane == 0) ctrl[0] = ok ? 0 : 1;
    }
    b.sync();
    int stop = ctrl[0];
    QPX_PROF(1)

    for (int it = 0; it < a.maxIter && !stop; ++it) {
        // T <- R ; R z'
        block_copy(b, Tm, Rg, tri(m));
        b.sync();
        QPX_PROF(2)
        block_symv_packed(b, Tm, m, vA, vB);
        b.sync();
        QPX_PROF(3)
        T mu = 0, feas = 0, resid = 0, szdot = 0;
        if (w0) {
            ld_slots<NS>(b, rzp, vB, m, T(0));
            T pri2 = 0;
            szdot = 0;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                if (i < m) {
                    const T rz = s[k] - c[k] - rzp[k];              // batch.py:99
                    pri2 = fma_(rz, rz, pri2);
                    szdot = fma_(s[k], z[k], szdot);
                    Tm[tri(i) + i] += s[k] / z[k];                   // + 1/d, d = z/s  batch.py:109,446
                }
            }
            pri2 = wave_sum(b, pri2);
            szdot = wave_sum(b, szdot);
            mu = abs_(szdot / mT);                                  // batch.py:102
            const T pri = sqrt_(pri2);
            const T dual = tau * sigz * g1n;
            feas = pri + dual;
            resid = feas + mT * mu;                                 // batch.py:107
            if (a.trace && lane == 0) {
                T* tr = a.trace + ((size_t)it * a.B + qp) * 3;
                tr[0] = pri; tr[1] = dual; tr[2] = mu;
            }
        }
        b.sync();
        QPX_PROF(4)
        ok = chol_packed(b, Tm, m, dinv);                            // factor_kkt  batch.py:110
        QPX_PROF(5)
        if (w0) {
            int stopf = 0;
            if (!ok) {
                st |= QPX_ST_KKT_BREAKDOWN;                          // `except: return best`
                stopf = 1;
            } else {
                iters = it + 1;
                const bool better = (it == 0) || (resid < bres);     // batch.py:118-139
                if (better) {
                    bres = resid; btau = tau; nnot = 0;
#pragma unroll
                    for (int k = 0; k < NS; ++k) { bz[k] = z[k]; bs[k] = s[k]; }
                } else if (a.stall_policy == 1 || (a.stall_policy == 2 && mT * mu < feas)) {
                    nnot += 1;
                } else {
                    nnot = 0;
                }
                if (a.stall_policy == 2 && it >= 1 && feas > T(2) * (T(1) - alpha_prev) * feas_prev)
                    floor_hit = 1;                                   // feasibility sits on round-off
                feas_prev = feas;
                if ((a.stall_policy != 0 && nnot >= a.notImprovedLim) || bres < a.eps || mu > T(1e32))
                    stopf = 1;                                       // batch.py:140
                if (a.stall_policy == 2 && floor_hit && mT * mu < T(1e-2) * feas) stopf = 1;
                if (!finite_(resid)) { stopf = 1; st |= QPX_ST_NONFINITE; }
            }
            if (!stopf) {
                // affine scaling direction (batch.py:145-151): rhs = c + R z = c + R z' + tau sigz R 1
                T dza[NS], dsa[NS], dz[NS], ds[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) dza[k] = c[k] + rzp[k] + tau * sigz * r1[k];
                solve_neg<NS>(b, Tm, dinv, m, dza);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int i = k * kWave + lane;
                    dsa[k] = (i < m) ? (-s[k] - dza[k] * s[k] / z[k]) : T(0);   // (-rs - dz)/d, rs = z
                    if (i >= m) dza[k] = T(0);
                }
                T al = step_to_boundary<NS>(b, z, dza, m);
                const T al2 = step_to_boundary<NS>(b, s, dsa, m);
                al = (al2 < al) ? al2 : al;
                al = (al < T(1)) ? al : T(1);                         // batch.py:160-162
                T t3 = 0;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int i = k * kWave + lane;
                    if (i < m) t3 = fma_(s[k] + al * dsa[k], z[k] + al * dza[k], t3);
                }
                t3 = wave_sum(b, t3);
                T sig = t3 / szdot;
                sig = sig * sig * sig;                                // batch.py:164-168
                // centering-corrector: rs = (-mu sig + ds_aff dz_aff)/s ; rhs = rs/d   batch.py:170-181
                T rs[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int i = k * kWave + lane;
                    rs[k] = (i < m) ? ((-mu * sig + dsa[k] * dza[k]) / s[k]) : T(0);
                    dz[k] = (i < m) ? (rs[k] * s[k] / z[k]) : T(0);
                }
                solve_neg<NS>(b, Tm, dinv, m, dz);                    // dz_cor
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int i = k * kWave + lane;
                    const T dsc = (i < m) ? ((-rs[k] - dz[k]) * s[k] / z[k]) : T(0);
                    dz[k] = (i < m) ? (dza[k] + dz[k]) : T(0);       // batch.py:189-192
                    ds[k] = dsa[k] + dsc;
                }
                al = step_to_boundary<NS>(b, z, dz, m);
                const T al3 = step_to_boundary<NS>(b, s, ds, m);
                al = (al3 < al) ? al3 : al;
                al = T(0.999) * al;
                al = (al < T(1)) ? al : T(1);                         // batch.py:193-195
                tau = (T(1) - al) * tau;
                alpha_prev = al;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int i = k * kWave + lane;
                    if (i < m) {
                        z[k] = fma_(al, dz[k], z[k]);                 // batch.py:200-203
                        s[k] = fma_(al, ds[k], s[k]);
                        vA[i] = z[k] - tau * sigz;
                    }
                }
            }
            if (lane == 0) ctrl[0] = stopf;
        }
        b.sync();
        stop = ctrl[0];
        QPX_PROF(6)
    }

    // ---- outputs: best iterate (batch.py:143,207): lam = z, slacks = s, zhat = x, nu = y
    if (w0) {
        if (iters >= a.maxIter && !(bres < a.eps)) st |= QPX_ST_MAXITER;
        if (!(bres <= T(1))) st |= QPX_ST_INACCURATE;               // batch.py:141,205
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) {
                a.lam[(size_t)qp * m + i] = bz[k];
                a.slack[(size_t)qp * m + i] = bs[k];
                vA[i] = bz[k] - btau * sigz;
            }
        }
        if (lane == 0) {
            a.iters[qp] = iters;
            a.status[qp] |= st;
            a.best_resid[qp] = bres;
        }
    }
    b.sync();
    // w = w0 - Zp z'   (one wave per row)
    {
        const T* Zp = F + lay.Zp;
        for (int k = b.wave(); k < n; k += b.nwaves()) {
            const T* zr = Zp + (size_t)k * m;
            T acc = 0;
            for (int j = lane; j < m; j += kWave) acc = fma_(zr[j], vA[j], acc);
            acc = wave_sum(b, acc);
            if (lane == 0) vB[k] = vW[k] - acc;
        }
    }
    const T* Lp = F + lay.L;
    const T* dLp = F + lay.dinvL;
    if (kLds) {
        block_copy(b, Tm, F + lay.L, tri(n));
        block_copy(b, dinv, F + lay.dinvL, (size_t)n);
        Lp = Tm;
        dLp = dinv;
    }
    b.sync();
    if (w0) {
        T x[NS];
        ld_slots<NS>(b, x, vB, n, T(0));
        trsv_bwd<NS>(b, Lp, dLp, n, x);                              // x = L^-T w
        st_slots<NS>(b, a.zhat + (size_t)qp * n, x, n);
        if (q > 0) {
            // nu = -L11^-T (ycoef + V z')
            const T* V = F + lay.V;
            const T* yc = vQ1;
            T y[NS];
#pragma unroll
            for (int sa = 0; sa < NS; ++sa) {
                y[sa] = T(0);
                for (int la = 0; la < kWave; ++la) {
                    const int r = sa * kWave + la;
                    if (r < q) {
                        T acc = 0;
                        for (int j = lane; j < m; j += kWave) acc = fma_(V[(size_t)r * m + j], vA[j], acc);
                        acc = wave_sum(b, acc);
                        if (lane == la) y[sa] = yc[r] + acc;
                    }
                }
            }
            trsv_bwd<NS>(b, F + lay.L11, F + lay.dinv11, q, y);
#pragma unroll
            for (int sa = 0; sa < NS; ++sa) y[sa] = -y[sa];
            st_slots<NS>(b, a.nu + (size_t)qp * q, y, q);
        }
    }
    QPX_PROF(7)
    QPX_PROF_DUMP(a.trace ? a.trace + (size_t)qp * 8 : (T*)nullptr, T)
}

// ------------------------------------------------------------------------------------------
// One factor_kkt + solve_kkt (batch.py:435-470, 349-372) on the factor blob; with kBackward
// the right-hand side and d are those of QPFunctionFn.backward and the six gradients are
// written (qp.py:148-177, per QP; the `.mean(0)` over a broadcast batch is left to the host).
template <class T, int NS, bool kLds, bool kBackward>
QPX_DEV void kkt_body(const Block& b, const KktArgs<T>& a, int qp, T* lds)
{
    const int n = a.n, m = a.m, q = a.q;
    const FacLayout lay = fac_layout(n, m, q, 0);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const size_t v = align4(max2(max2((size_t)n, (size_t)m), (size_t)q));
    T* dinv = lds;
    T* vt = dinv + v;     // t = L^-1 rx              (n)
    T* vh = vt + v;       // Zp^T t                   (m)
    T* vyt = vh + v;      // Yh^T t                   (q)
    T* vdz = vyt + v;     // dz                       (m)
    T* vw = vdz + v;      // w / dx                   (n)
    T* vrho = vw + v;     // rho = L11^-1 ry, later Yt - rho (q)
    T* vdy = vrho + v;    // dy                       (q)
    int* ctrl = reinterpret_cast<int*>(vdy + v);
    T* Tm = kLds ? (vdy + v + 4) : (F + lay.T);
    (void)ctrl;

    const int lane = b.lane();
    const bool w0 = b.wave() == 0;
    const T* Zp = F + lay.Zp;
    const T* Yh = F + lay.Yh;
    const T* V = F + lay.V;
    const T* rxg = kBackward ? (a.dl_dz + (size_t)qp * n) : (a.rx ? a.rx + (size_t)qp * n : nullptr);
    const T* rsg = (!kBackward && a.rs) ? a.rs + (size_t)qp * m : nullptr;
    const T* rzg = (!kBackward && a.rz) ? a.rz + (size_t)qp * m : nullptr;
    const T* ryg = (!kBackward && a.ry && q > 0) ? a.ry + (size_t)qp * q : nullptr;

    // 1. t = L^-1 rx
    const T* Lp = F + lay.L;
    const T* dLp = F + lay.dinvL;
    if (kLds) {
        block_copy(b, Tm, F + lay.L, tri(n));
        block_copy(b, dinv, F + lay.dinvL, (size_t)n);
        Lp = Tm;
        dLp = dinv;
    }
    b.sync();
    if (w0) {
        T x[NS];
        if (rxg) ld_slots<NS>(b, x, rxg, n, T(0));
        else {
#pragma unroll
            for (int k = 0; k < NS; ++k) x[k] = T(0);
        }
        trsv_fwd<NS>(b, Lp, dLp, n, x);
        st_slots<NS>(b, vt, x, n);
        // rho = L11^-1 ry
        if (q > 0) {
            T r[NS];
            if (ryg) ld_slots<NS>(b, r, ryg, q, T(0));
            else {
#pragma unroll
                for (int k = 0; k < NS; ++k) r[k] = T(0);
            }
            trsv_fwd<NS>(b, F + lay.L11, F + lay.dinv11, q, r);
            st_slots<NS>(b, vrho, r, q);
        }
    }
    b.sync();
    // 2. h = Zp^T t ; Yt = Yh^T t ; T <- R
    for (int j = b.tid; j < m; j += b.nt) {
        T acc = 0;
        for (int k = 0; k < n; ++k) acc = fma_(Zp[(size_t)k * m + j], vt[k], acc);
        vh[j] = acc;
    }
    for (int r = b.tid; r < q; r += b.nt) {
        T acc = 0;
        for (int k = 0; k < n; ++k) acc = fma_(Yh[(size_t)k * q + r], vt[k], acc);
        vyt[r] = acc;
    }
    block_copy(b, Tm, F + lay.R, tri(m));
    b.sync();
    // 3. T = R + diag(1/d) ; factor
    T d[NS];
    if (w0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            d[k] = T(1);
            if (i < m) {
                if (kBackward) {
                    const T l = a.lam[(size_t)qp * m + i], sl = a.slack[(size_t)qp * m + i];
                    d[k] = ((l < T(1e-8)) ? T(1e-8) : l) / ((sl < T(1e-8)) ? T(1e-8) : sl);   // qp.py:148
                } else {
                    d[k] = a.d[(size_t)qp * m + i];
                }
                Tm[tri(i) + i] += T(1) / d[k];
            }
        }
    }
    const bool ok = chol_packed(b, Tm, m, dinv);
    if (!ok && b.tid == 0 && a.status) a.status[qp] |= QPX_ST_KKT_BREAKDOWN;
    // 4. dz = -T^-1 (h + V^T rho + rs/d - rz) ; ds = (-rs - dz)/d ; dy = L11^-T (rho - Yt - V dz)
    if (w0) {
        T x[NS], rs[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            rs[k] = (rsg && i < m) ? rsg[i] : T(0);
            T acc = T(0);
            if (i < m) {
                acc = vh[i] + rs[k] / d[k] - ((rzg) ? rzg[i] : T(0));
                for (int r = 0; r < q; ++r) acc = fma_(V[(size_t)r * m + i], vrho[r], acc);
            }
            x[k] = ok ? acc : T(0);
        }
        if (ok) solve_neg<NS>(b, Tm, dinv, m, x);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) {
                vdz[i] = x[k];
                const T dsv = (-rs[k] - x[k]) / d[k];
                if (kBackward) {
                    if (a.dh) a.dh[(size_t)qp * m + i] = -x[k];             // qp.py:161
                    if (a.dz) a.dz[(size_t)qp * m + i] = x[k];
                } else {
                    a.dz[(size_t)qp * m + i] = x[k];
                    a.ds[(size_t)qp * m + i] = dsv;
                }
            }
        }
        if (q > 0) {
            T y[NS];
#pragma unroll
            for (int sa = 0; sa < NS; ++sa) {
                y[sa] = T(0);
                for (int la = 0; la < kWave; ++la) {
                    const int r = sa * kWave + la;
                    if (r < q) {
                        T acc = 0;
                        for (int j = lane; j < m; j += kWave) acc = fma_(V[(size_t)r * m + j], vdz[j], acc);
                        acc = wave_sum(b, acc);
                        if (lane == la) y[sa] = vrho[r] - vyt[r] - acc;
                    }
                }
            }
            trsv_bwd<NS>(b, F + lay.L11, F + lay.dinv11, q, y);
            st_slots<NS>(b, vdy, y, q);
#pragma unroll
            for (int sa = 0; sa < NS; ++sa) {
                const int r = sa * kWave + lane;
                if (r < q) {
                    if (kBackward) {
                        if (a.db) a.db[(size_t)qp * q + r] = -y[sa];        // qp.py:166
                        if (a.dy) a.dy[(size_t)qp * q + r] = y[sa];
                    } else a.dy[(size_t)qp * q + r] = y[sa];
                    vrho[r] = vyt[r] - vrho[r];                             // Yt - rho
                }
            }
        }
    }
    b.sync();
    // 5. w = t - Yh (Yt - rho) + Zp dz ; 6. dx = -L^-T w
    for (int k = b.wave(); k < n; k += b.nwaves()) {
        const T* zr = Zp + (size_t)k * m;
        T acc = 0;
        for (int j = lane; j < m; j += kWave) acc = fma_(zr[j], vdz[j], acc);
        for (int r = lane; r < q; r += kWave) acc = fma_(-Yh[(size_t)k * q + r], vrho[r], acc);
        acc = wave_sum(b, acc);
        if (lane == 0) vw[k] = vt[k] + acc;
    }
    if (kLds) {
        block_copy(b, Tm, F + lay.L, tri(n));
        block_copy(b, dinv, F + lay.dinvL, (size_t)n);
    }
    b.sync();
    if (w0) {
        T x[NS];
        ld_slots<NS>(b, x, vw, n, T(0));
        trsv_bwd<NS>(b, Lp, dLp, n, x);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < n) {
                vw[i] = -x[k];
                if (kBackward) {
                    if (a.dp) a.dp[(size_t)qp * n + i] = -x[k];             // qp.py:157
                    if (a.dx) a.dx[(size_t)qp * n + i] = -x[k];
                } else a.dx[(size_t)qp * n + i] = -x[k];
            }
        }
    }
    if (kBackward) {
        b.sync();
        // gradient outer products (qp.py:158-173); vt <- zhat
        const T* zh = a.zhat + (size_t)qp * n;
        for (int k = b.tid; k < n; k += b.nt) vt[k] = zh[k];
        for (int j = b.tid; j < m; j += b.nt) vh[j] = a.lam[(size_t)qp * m + j];
        for (int r = b.tid; r < q; r += b.nt) vyt[r] = a.nu[(size_t)qp * q + r];
        b.sync();
        T* dQ = a.dQ ? a.dQ + (size_t)qp * n * n : nullptr;        // NULL = gradient not wanted
        for (int idx = b.tid; dQ && idx < n * n; idx += b.nt) {
            const int r = idx / n, cidx = idx - r * n;
            dQ[idx] = T(0.5) * (vw[r] * vt[cidx] + vt[r] * vw[cidx]);
        }
        T* dG = a.dG ? a.dG + (size_t)qp * m * n : nullptr;
        for (int idx = b.tid; dG && idx < m * n; idx += b.nt) {
            const int r = idx / n, cidx = idx - r * n;
            dG[idx] = vdz[r] * vt[cidx] + vh[r] * vw[cidx];
        }
        if (q > 0 && a.dA) {
            T* dA = a.dA + (size_t)qp * q * n;
            for (int idx = b.tid; idx < q * n; idx += b.nt) {
                const int r = idx / n, cidx = idx - r * n;
                dA[idx] = vdy[r] * vt[cidx] + vyt[r] * vw[cidx];
            }
        }
    }
}

}  // namespace qpx
