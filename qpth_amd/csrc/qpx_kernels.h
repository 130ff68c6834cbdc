// qpx_kernels.h -- what every kernel family shares: wave-level reductions, the one-wave vector slots, the widening
// loads of QPX_F32_WIDE and the argument blocks of the launches (plain data, passed by value).
//
// (Rounds 1-4 also kept the round-1 WORKGROUP kernels here -- prefactor_body / ipm_body / kkt_body: one 256-thread workgroup
// per QP, packed Cholesky factors and substitutions in LDS or in the blob -- reachable through knob 1 only since the
// large-QP family took every size beyond the thread-grid / tile kernels in round 4 (3-6 x faster there, profiles/archive/r04b).
// Deleted in round 5; libqpx_hip_r04.so, archived beside the product, still holds them for A/Bs.)
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

QPX_LAYOUT_HD size_t max2(size_t a, size_t b) { return a > b ? a : b; }

// ------------------------------------------------------------------------------------------
// step length to the boundary for one QP (get_step, batch.py:210-213, as it behaves for a batch of one): min over blocking
// entries of -v/dv, or 1 when nothing blocks -- from the reciprocals rv = 1/v the caller already holds: the minimum of -v/dv
// is the reciprocal of the maximum of -dv/v, so the per-entry divisions become multiplications and
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

}  // namespace qpx
