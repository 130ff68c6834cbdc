// qpx_grid.h -- the "thread-grid" kernels: a GS x GS grid of threads (GS = 16: one 256-thread
// workgroup; GS = 8: one wave64) owns one QP and keeps its symmetric work matrix REGISTER-RESIDENT
// in a 2-D cyclic layout: thread (a, b) = (tid % GS, tid / GS) holds the elements
// (GS*li + a, GS*lj + b), li >= lj, of the lower block triangle as NBL(NBL+1)/2 registers
// (diagonal GS x GS blocks are held in full).
//
// The design follows what the MI355X micro-benchmarks (scripts/ubench.py, profiles/) say:
// dependent f64 FMAs issue back to back (4.7 ticks), an LDS publish->consume round trip is ~85
// ticks, rcp ~42 -- but a triangular substitution step costs a serial readlane->fma chain per
// column, and four of them per IPM iteration cost more than the factorisation itself.  So the
// factorisation here produces the INVERSE of the unit-lower factor in place:
//
//   ldl_inv:  T = L~ D L~^T by right-looking rank-1 updates; the registers of the columns that
//             have been eliminated are reused for W~ = L~^-1, which is built by the SAME rank-1
//             update (row i > k of W~ gets -l~_ik * row k of W~).  One published vector and one
//             barrier per column; no substitutions afterwards:
//   solve:    x = -T^-1 r = -W~^T D^-1 W~ r  -- two triangular mat-vecs, all threads busy.
//
// Accuracy: the explicit unit-lower inverse loses nothing measurable against substitution on the
// IPM's matrices (z* error vs the reference 2.7e-12 max at C2 in f64; a full explicit T^-1 by
// Gauss-Jordan would lose 7 digits) -- see DESIGN.md section 7.
#pragma once
#include "qpx_kernels.h"

namespace qpx {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a compile-time constant
template <int I0, int N, class F> QPX_DEV void static_for_(F&& f)
{
    if constexpr (I0 < N) {
        f(std::integral_constant<int, I0>{});
        static_for_<I0 + 1, N>(f);
    }
}
template <int N, class F> QPX_DEV void static_for(F&& f) { static_for_<0, N>(f); }

constexpr int gtri(int x) { return x * (x + 1) / 2; }
constexpr int gidx(int li, int lj) { return li * (li + 1) / 2 + lj; }   // li >= lj

template <int GS> struct GridPos {
    int tid, a, b;
    static constexpr int NT = GS * GS;
    QPX_DEV explicit GridPos(const Block& blk) : tid(blk.tid), a(blk.tid % GS), b(blk.tid / GS) {}
    // (the tile kernels' chain-wave form assigns wave roles here and hands the loop's vector work to its chain wave)
    QPX_DEV void assign(const Block&, int*) {}
    QPX_DEV bool lead(const Block& blk) const { return blk.wave() == 0; }
    // all threads of the grid reach this point; LDS writes before it are visible after it
    static QPX_DEV void sync(const Block& blk)
    {
        if (GS == 8) blk.wave_sync();
        else blk.sync();
    }
};

// element count of the grid-layout copy of a symmetric matrix of NBL x NBL blocks
QPX_LAYOUT_HD size_t grid_elems(int gs, int nbl) { return (size_t)(nbl * (nbl + 1) / 2) * gs * gs; }

// E <- matrix stored in grid layout: entry [gidx(li,lj)*GS*GS + tid]
template <class T, int GS, int NBL>
QPX_DEV void grid_load(const Block& blk, T (&E)[gtri(NBL)], const T* Rg)
{
    constexpr int NT = GS * GS;
    if (GS == 8) {
        const GlobalRows<T> rows(Rg, gtri(NBL) * NT, blk.tid);
#pragma unroll
        for (int e = 0; e < gtri(NBL); ++e) E[e] = rows.row(e);
    } else {
#pragma unroll
        for (int e = 0; e < gtri(NBL); ++e) E[e] = Rg[(size_t)e * NT + blk.tid];
    }
}

// Sum NBL per-thread partials over one grid axis through LDS and (accumulate) into out[GS*l + r].
// kOverB: sum over b (result indexed by a: "row sums"), else over a ("column sums").
template <class T, int GS, int NBL, bool kOverB, bool kAccumulate>
QPX_DEV void grid_reduce(const Block& blk, const GridPos<GS>& g, const T (&part)[NBL], T* red, T* out)
{
    constexpr int NT = GS * GS;
    const int r = kOverB ? g.a : g.b, c = kOverB ? g.b : g.a;
#pragma unroll
    for (int l = 0; l < NBL; ++l) red[(l * GS + r) * GS + c] = part[l];
    GridPos<GS>::sync(blk);
    for (int idx = g.tid; idx < NBL * GS; idx += NT) {
        const T* p = red + idx * GS;
        T s = T(0);
#pragma unroll
        for (int k = 0; k < GS; ++k) s += p[k];
        out[idx] = kAccumulate ? (out[idx] + s) : s;
    }
    GridPos<GS>::sync(blk);
}

// vout = S vin for the symmetric matrix in E (lower blocks, diagonal blocks in full).
template <class T, int GS, int NBL>
QPX_DEV void grid_symv(const Block& blk, const GridPos<GS>& g, const T (&E)[gtri(NBL)], const T* vin, T* vout, T* red)
{
    T racc[NBL], cacc[NBL];
#pragma unroll
    for (int l = 0; l < NBL; ++l) racc[l] = cacc[l] = T(0);
#pragma unroll
    for (int li = 0; li < NBL; ++li) {
        const T vri = vin[GS * li + g.a];
#pragma unroll
        for (int lj = 0; lj <= li; ++lj) {
            const T t = E[gidx(li, lj)];
            racc[li] = fma_(t, vin[GS * lj + g.b], racc[li]);
            if (li != lj) cacc[lj] = fma_(t, vri, cacc[lj]);
        }
    }
    grid_reduce<T, GS, NBL, true, false>(blk, g, racc, red, vout);
    grid_reduce<T, GS, NBL, false, true>(blk, g, cacc, red, vout);
}

template <class T, int GS, int NBL>
QPX_DEV void grid_add_diag(const GridPos<GS>& g, T (&E)[gtri(NBL)], const T* vd)
{
    if (g.a == g.b) {
#pragma unroll
        for (int l = 0; l < NBL; ++l) E[gidx(l, l)] += vd[GS * l + g.a];
    }
}

// One column step of ldl_inv for column k = GS*KB + ka.  `vec` (double buffered by the parity of
// k) receives: rows i > k: c_ik (un-scaled column k of the Schur complement), cols j < k:
// W~_kj (row k of the inverse factor, final), entry k: 0; `dsl[parity]` the pivot d_k.
template <class T, int GS, int NBL, int KB>
QPX_DEV bool grid_ldl_inv_step(const Block& blk, const GridPos<GS>& g, T (&E)[gtri(NBL)], T* vec2, T* dsl, T* rd, int ka)
{
    constexpr int M = GS * NBL;
    const int k = GS * KB + ka;
    T* vec = vec2 + (k & 1) * M;
    // ---- publish
    if (g.b == ka) {
#pragma unroll
        for (int li = KB; li < NBL; ++li) {
            const int i = GS * li + g.a;
            if (i > k) vec[i] = E[gidx(li, KB)];
        }
    }
    if (g.a == ka) {
#pragma unroll
        for (int lj = 0; lj <= KB; ++lj) {
            const int j = GS * lj + g.b;
            if (j < k) vec[j] = E[gidx(KB, lj)];
        }
        if (g.b == ka) {
            const T d = E[gidx(KB, KB)];
            vec[k] = T(0);
            dsl[k & 1] = d;
        }
    }
    GridPos<GS>::sync(blk);
    // ---- consume
    const T dk = dsl[k & 1];
    if (!(dk > T(0)) || !finite_(dk)) return false;
    const T r = rcp_(dk);
    if (g.tid == 0) rd[k] = r;
    T lrow[NBL];
#pragma unroll
    for (int li = KB; li < NBL; ++li) {
        const T v = vec[GS * li + g.a];
        lrow[li] = (li > KB || g.a > ka) ? v * r : T(0);      // rows <= k take no part
    }
#pragma unroll
    for (int lj = 0; lj < NBL; ++lj) {
        const T y = vec[GS * lj + g.b];
#pragma unroll
        for (int li = (lj > KB ? lj : KB); li < NBL; ++li) {
            T e = fma_(-lrow[li], y, E[gidx(li, lj)]);
            // column k itself becomes the new column of W~: W~_ik = -l~_ik, assigned exactly (forming
            // it as c_ik - l~_ik (d_k + 1) would cancel with relative error eps * d_k)
            if (lj == KB && g.b == ka && (li > KB || g.a > ka)) e = -lrow[li];
            E[gidx(li, lj)] = e;
        }
    }
    return true;
}

template <class T, int GS, int NBL, int KB> struct GridLdlBlocks {
    static QPX_DEV bool run(const Block& blk, const GridPos<GS>& g, T (&E)[gtri(NBL)], T* vec2, T* dsl, T* rd, int m)
    {
        if (GS * KB >= m) return true;               // the pad is the identity: nothing to eliminate
        const int kend = (m - GS * KB < GS) ? (m - GS * KB) : GS;
#pragma unroll 1
        for (int ka = 0; ka < kend; ++ka)
            if (!grid_ldl_inv_step<T, GS, NBL, KB>(blk, g, E, vec2, dsl, rd, ka)) return false;
        return GridLdlBlocks<T, GS, NBL, KB + 1>::run(blk, g, E, vec2, dsl, rd, m);
    }
};
template <class T, int GS, int NBL> struct GridLdlBlocks<T, GS, NBL, NBL> {
    static QPX_DEV bool run(const Block&, const GridPos<GS>&, T (&)[gtri(NBL)], T*, T*, T*, int) { return true; }
};

// E: T (SPD, order m padded with the identity) -> strictly-lower part: W~ = L~^-1, diagonal: d_k.
// rd[k] = 1/d_k.  vec2: 2*GS*NBL elements of LDS, dsl: 2.  Uniform return value.
template <class T, int GS, int NBL>
QPX_DEV bool grid_ldl_inv(const Block& blk, const GridPos<GS>& g, T (&E)[gtri(NBL)], T* vec2, T* dsl, T* rd, int m)
{
    const bool ok = GridLdlBlocks<T, GS, NBL, 0>::run(blk, g, E, vec2, dsl, rd, m);
    GridPos<GS>::sync(blk);
    return ok;
}

// vout = -T^-1 vin = -W~^T D^-1 W~ vin with W~ (unit lower) in E; vin, vout, tmp: LDS vectors of
// length GS*NBL (pad entries of vin must be zero); entries >= m of rd are not read.
template <class T, int GS, int NBL>
QPX_DEV void grid_solve_neg(const Block& blk, const GridPos<GS>& g, const T (&E)[gtri(NBL)], const T* rd, int m,
                            const T* vin, T* vout, T* tmp, T* red)
{
    constexpr int NT = GS * GS;
    T acc[NBL];
    // y = W~ vin   (row sums over j < i, plus the unit diagonal)
#pragma unroll
    for (int l = 0; l < NBL; ++l) acc[l] = T(0);
#pragma unroll
    for (int li = 0; li < NBL; ++li)
#pragma unroll
        for (int lj = 0; lj <= li; ++lj) {
            const T w = (lj < li || g.b < g.a) ? E[gidx(li, lj)] : T(0);
            acc[li] = fma_(w, vin[GS * lj + g.b], acc[li]);
        }
    grid_reduce<T, GS, NBL, true, false>(blk, g, acc, red, tmp);
    for (int i = g.tid; i < GS * NBL; i += NT) tmp[i] = (i < m) ? (tmp[i] + vin[i]) * rd[i] : T(0);   // u = D^-1 y
    GridPos<GS>::sync(blk);
    // x = W~^T u   (column sums over i > j, plus the unit diagonal)
#pragma unroll
    for (int l = 0; l < NBL; ++l) acc[l] = T(0);
#pragma unroll
    for (int li = 0; li < NBL; ++li) {
        const T ui = tmp[GS * li + g.a];
#pragma unroll
        for (int lj = 0; lj <= li; ++lj) {
            const T w = (lj < li || g.b < g.a) ? E[gidx(li, lj)] : T(0);
            acc[lj] = fma_(w, ui, acc[lj]);
        }
    }
    grid_reduce<T, GS, NBL, false, false>(blk, g, acc, red, vout);
    for (int i = g.tid; i < GS * NBL; i += NT) vout[i] = -(vout[i] + tmp[i]);
    GridPos<GS>::sync(blk);
}


// The register-resident matrix operations of the PDIPM loop as a policy, so that the loop body
// (ipm_loop_body) is written once for every register layout (thread grid here, MFMA tiles in
// qpx_tile.h).  MP = padded order, NT = threads per QP, scratch = LDS elements the operations use.
template <class T, int GS, int NBL> struct GridMat {
    static constexpr int NT = GS * GS, MP = GS * NBL;
    static constexpr bool kAhead = false;      // (TileMat's chain-wave form: the loop's phases re-ordered around a wave that runs ahead)
    using Pos = GridPos<GS>;
    template <class F> static QPX_DEV void with_role(const Pos& p, F&& f) { f(p); }      // (see TileMat::with_role)
    struct Regs { T e[gtri(NBL)]; };
    QPX_LAYOUT_HD static size_t scratch_elems() { return 2 * (size_t)MP + 4 + (size_t)NBL * GS * GS; }
    static QPX_DEV void sync(const Block& blk) { GridPos<GS>::sync(blk); }
    static QPX_DEV const T* image(const T* F, const FacLayout& lay) { return F + (GS == 8 ? lay.Rw : lay.Rg); }
    static QPX_DEV void load(const Block& blk, const Pos&, Regs& E, const T* img) { grid_load<T, GS, NBL>(blk, E.e, img); }
    template <bool kLeadOnly = false>
    static QPX_DEV void symv(const Block& blk, const Pos& g, const Regs& E, const T* vin, T* vout, T* scratch)
    {
        grid_symv<T, GS, NBL>(blk, g, E.e, vin, vout, scratch + 2 * MP + 4);
    }
    static QPX_DEV void add_diag(const Pos& g, Regs& E, const T* vd) { grid_add_diag<T, GS, NBL>(g, E.e, vd); }
    static QPX_DEV bool ldl_inv(const Block& blk, const Pos& g, Regs& E, T* scratch, T* rd, int m)
    {
        return grid_ldl_inv<T, GS, NBL>(blk, g, E.e, scratch, scratch + 2 * MP, rd, m);
    }
    template <bool kLeadOnly = false>
    static QPX_DEV void solve_neg(const Block& blk, const Pos& g, const Regs& E, const T* rd, int m, const T* vin,
                                  T* vout, T* tmp, T* scratch)
    {
        grid_solve_neg<T, GS, NBL>(blk, g, E.e, rd, m, vin, vout, tmp, scratch + 2 * MP + 4);
    }
};

// LDS elements of the loop kernel: 17 vectors of v = 64 NS >= max(n, MP, q), 16 scalars + 8 control
// words, scratch
QPX_LAYOUT_HD size_t lds_elems_ipm_loop(size_t mp, size_t scratch, int n, int q)
{
    const size_t d = max2(max2((size_t)n, mp), (size_t)q);
    const size_t v = d <= 64 ? 64 : (d <= 128 ? 128 : (d <= 256 ? 256 : 512));    // 64 NS, see ipm_loop_body
    return 17 * v + 24 + scratch;
}
QPX_LAYOUT_HD size_t lds_elems_ipm_grid(int gs, int nbl, int n, int q)
{
    const size_t mg = (size_t)gs * nbl;
    return lds_elems_ipm_loop(mg, 2 * mg + 4 + (size_t)nbl * gs * gs, n, q);
}

// out[c] (op)= sum_r Mat[r][c] * vec[r]   -- thread per column c: the loads of a row are
// coalesced across threads and independent across r (deep memory pipeline, no reduction).
template <class T, int MODE /*0: =, 1: +=, 2: -=*/, class Acc = T /* accumulate in (e.g. double for float data) */>
QPX_DEV void block_matTvec(const Block& blk, T* out, const T* Mat, const T* vec, int rows, int cols)
{
    for (int c = blk.tid; c < cols; c += blk.nt) {
        // eight independent loads in flight per thread: the rows come from L2 / HBM (~800 ticks each way)
        Acc a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
        const T* col = Mat + c;
        int r = 0;
        for (; r + 8 <= rows; r += 8) {
            const T m0 = col[(size_t)r * cols], m1 = col[(size_t)(r + 1) * cols], m2 = col[(size_t)(r + 2) * cols];
            const T m3 = col[(size_t)(r + 3) * cols], m4 = col[(size_t)(r + 4) * cols], m5 = col[(size_t)(r + 5) * cols];
            const T m6 = col[(size_t)(r + 6) * cols], m7 = col[(size_t)(r + 7) * cols];
            a0 = fma_((Acc)m0, (Acc)vec[r], a0);
            a1 = fma_((Acc)m1, (Acc)vec[r + 1], a1);
            a2 = fma_((Acc)m2, (Acc)vec[r + 2], a2);
            a3 = fma_((Acc)m3, (Acc)vec[r + 3], a3);
            a4 = fma_((Acc)m4, (Acc)vec[r + 4], a4);
            a5 = fma_((Acc)m5, (Acc)vec[r + 5], a5);
            a6 = fma_((Acc)m6, (Acc)vec[r + 6], a6);
            a7 = fma_((Acc)m7, (Acc)vec[r + 7], a7);
        }
        for (; r < rows; ++r) a0 = fma_((Acc)col[(size_t)r * cols], (Acc)vec[r], a0);
        const Acc sum = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
        out[c] = MODE == 0 ? (T)sum : (MODE == 1 ? (T)((Acc)out[c] + sum) : (T)((Acc)out[c] - sum));
    }
}

// out[r] (op)= sum_c Mat[r][c] * vec[c]  -- row dots: a wave takes RB rows at a time, its lanes stride over the
// columns (coalesced), RB independent loads in flight per lane, then RB wave reductions.
template <class T, int MODE /*0: =, 1: +=, 2: -=*/, class Acc = T>
QPX_DEV void block_matvec(const Block& blk, T* out, const T* Mat, const T* vec, int rows, int cols)
{
    constexpr int RB = 4;
    const int lane = blk.lane(), w = blk.uniform(blk.wave()), nw = blk.nwaves();
    for (int r0 = w * RB; r0 < rows; r0 += nw * RB) {
        Acc acc[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) acc[u] = Acc(0);
        for (int c = lane; c < cols; c += kWave) {
            const Acc x = (Acc)vec[c];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = (r0 + u < rows) ? r0 + u : rows - 1;      // clamped: loads stay unconditional
                acc[u] = fma_((Acc)Mat[(size_t)r * cols + c], x, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) acc[u] = wave_sum(blk, acc[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < RB; ++u)
                if (r0 + u < rows)
                    out[r0 + u] = MODE == 0 ? (T)acc[u] : (MODE == 1 ? (T)((Acc)out[r0 + u] + acc[u]) : (T)((Acc)out[r0 + u] - acc[u]));
        }
    }
}

// The same row dots on tiles of 16 rows: lane (g, c) of a wave takes rows r0 + g + 4 r (r = 0 .. 3) and the columns
// c, c + 16, ... -- 16 lanes read 128 consecutive bytes of a row, sixteen independent loads in flight per lane -- and the
// sums over a row's sixteen lanes are four DPP steps per register instead of one wave reduction per row (the
// round-2 form above spent most of its time in those: the loop kernel's epilogue, 25 000 cycles per QP).
template <class T, int MODE /*0: =, 1: +=, 2: -=*/, class Acc = T>
QPX_DEV void block_matvec16(const Block& blk, T* out, const T* Mat, const T* vec, int rows, int cols)
{
    const int lane = blk.lane(), g = lane >> 4, c = lane & 15, w = blk.uniform(blk.wave()), nw = blk.nwaves();
    for (int r0 = 16 * w; r0 < rows; r0 += 16 * nw) {
        Acc acc[4] = {Acc(0), Acc(0), Acc(0), Acc(0)};
        const T* rowp[4];
        bool live[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + g + 4 * r;
            live[r] = row < rows;
            rowp[r] = Mat + (size_t)(live[r] ? row : rows - 1) * cols;       // clamped: loads stay unconditional
        }
        for (int c0 = 0; c0 < cols; c0 += 64) {
            T mv[4][4];
            Acc xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = c0 + 16 * j + c;
                const bool in = col < cols;
                xv[j] = in ? (Acc)vec[col] : Acc(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) mv[j][r] = rowp[r][in ? col : cols - 1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fma_((Acc)mv[j][r], xv[j], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Acc v = acc[r];
            v += blk.template xor16<1>(v);
            v += blk.template xor16<2>(v);
            v += blk.template xor16<7>(v);
            v += blk.template xor16<15>(v);
            if (c == 0 && live[r]) {
                const int row = r0 + g + 4 * r;
                out[row] = MODE == 0 ? (T)v : (MODE == 1 ? (T)((Acc)out[row] + v) : (T)((Acc)out[row] - v));
            }
        }
    }
}

// Two column-parallel products with one vector at once: outA[c] (opA)= sum_r MatA[r][c] vec[r] (c < colsA) and
// outB[c] (opB)= sum_r MatB[r][c] vec[r] (c < colsB): the columns of both matrices are dealt over ALL threads (one
// product alone keeps 100 of 256 threads busy at C2) and sixteen loads are in flight per thread.
template <class T, int MODEA, int MODEB, class Acc = T, int D = 16 /* loads in flight per thread */>
QPX_DEV void block_matTvec2(const Block& blk, T* outA, const T* MatA, int colsA, T* outB, const T* MatB, int colsB,
                            const T* vec, int rows)
{
    const int colsAP = (colsA + 63) & ~63;                      // B's columns start at a wave boundary
    for (int cc = blk.tid; cc < colsAP + colsB; cc += blk.nt) {
        const bool isA = cc < colsAP;
        const int c = isA ? cc : cc - colsAP, cols = isA ? colsA : colsB;
        if (isA && c >= colsA) continue;
        const T* col = (isA ? MatA : MatB) + c;
        Acc a[D];
#pragma unroll
        for (int u = 0; u < D; ++u) a[u] = Acc(0);
        // (the last, partial batch too is sixteen unconditional loads -- of the last row again, times zero: a tail loop
        // of single loads was four more round trips to memory at rows = 100)
        for (int r = 0; r < rows; r += D) {
            T mv[D];
#pragma unroll
            for (int u = 0; u < D; ++u) mv[u] = col[(size_t)(r + u < rows ? r + u : rows - 1) * cols];
#pragma unroll
            for (int u = 0; u < D; ++u) a[u] = fma_((Acc)mv[u], r + u < rows ? (Acc)vec[r + u] : Acc(0), a[u]);
        }
        Acc sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        if constexpr (D == 16) sum += ((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15]));
        T* o = (isA ? outA : outB) + c;
        const int mode = isA ? MODEA : MODEB;
        *o = mode == 0 ? (T)sum : (mode == 1 ? (T)((Acc)*o + sum) : (T)((Acc)*o - sum));
    }
}

// One pivot of the symmetric sweep operator on the register-resident matrix (pivot k = 16*KB + ka):
//   E_ij -= v_i v_j / d for i, j != k,   E_ik = E_ki = v_i / d,   E_kk = -1/d,   v = column k.
// Pivots k < npos must be positive (SPD Q), the others negative (-A Q^-1 A^T).
template <class T, int NBL, int KB>
QPX_DEV int sweep_step(const Block& blk, const GridPos<16>& g, T (&E)[gtri(NBL)], T* vec2, T* dsl, int ka, int npos)
{
    constexpr int GS = 16, MA = GS * NBL;
    const int k = GS * KB + ka;
    T* vec = vec2 + (k & 1) * MA;
    if (g.b == ka) {
#pragma unroll
        for (int li = KB; li < NBL; ++li) {
            const int i = GS * li + g.a;
            if (i > k) vec[i] = E[gidx(li, KB)];
        }
    }
    if (g.a == ka) {
#pragma unroll
        for (int lj = 0; lj <= KB; ++lj) {
            const int j = GS * lj + g.b;
            if (j < k) vec[j] = E[gidx(KB, lj)];
        }
        if (g.b == ka) {
            const T d = E[gidx(KB, KB)];
            vec[k] = T(0);
            dsl[k & 1] = d;
        }
    }
    GridPos<GS>::sync(blk);
    const T d = dsl[k & 1];
    const bool okp = (k < npos) ? (d > T(0)) : (d < T(0));
    if (!okp || !finite_(d)) return (k < npos) ? QPX_ST_Q_NOT_SPD : QPX_ST_A_RANK;
    const T r = rcp_(d);
    T vr[NBL];
#pragma unroll
    for (int l = 0; l < NBL; ++l) vr[l] = vec[GS * l + g.a] * r;
#pragma unroll
    for (int lj = 0; lj < NBL; ++lj) {
        const T y = vec[GS * lj + g.b];
#pragma unroll
        for (int li = lj; li < NBL; ++li) {
            T e = fma_(-vr[li], y, E[gidx(li, lj)]);
            // row k and column k of the swept matrix are v / d exactly (v_k = 0 leaves them untouched
            // by the rank-1 update; forming them through it would cancel with relative error eps*d)
            if (lj == KB && g.b == ka) e = vr[li];                                   // (i, k), any i
            if (li == KB && g.a == ka) e = vec[GS * lj + g.b] * r;                   // (k, j)
            E[gidx(li, lj)] = e;
        }
    }
    if (g.a == ka && g.b == ka) E[gidx(KB, KB)] = -r;
    return 0;
}

// FOUR pivots at once (k0 .. k0 + 3, k0 = 16 KB + 4 gq): the sweep of the pivot BLOCK P = E(k0 .. k0+3, k0 .. k0+3),
//   E_ij -= (C P^-1 C^T)_ij  for i, j outside the block,   E_i,P = (C P^-1)_i,   E_PP = -P^-1,     C = columns k0 .. k0+3
// -- exactly what four rank-1 steps compose to, with two barriers and one serial chain (publish -> barrier -> reciprocals)
// per four pivots instead of four: one pivot of the rank-1 form cost ~2 400 cycles with two QPs per CU, against ~830
// for its 26 LDS reads and 91 FMAs per thread (profiles/archive/r03a: the sweep was 74 % of the pre-factorisation).
//   A  the owners publish the four columns (from the LOWER triangle only, as the rank-1 step: column k0+t below its
//      pivot, row k0+t left of it)                                                              -- barrier 1
//   B  every thread sweeps the 4 x 4 block P in registers (-> -P^-1, the same arithmetic in every thread, so the
//      failure decision is uniform) and thread i < MA forms row i of U = C P^-1 -> LDS          -- barrier 2
//   C  four rank-1 updates E -= U_t C_t^T back to back (no barrier between them: the LDS reads of one run beside the
//      FMAs of the previous), rows and columns of the block masked out of them; then the block's own rows and columns
//      are ASSIGNED (U, -P^-1), never formed through the update (see sweep_step).
// C is double-buffered by the parity of gq (a thread may publish group gq + 1 while another still reads group gq).
template <class T, int NBL, int KB>
QPX_DEV int sweep_group4(const Block& blk, const GridPos<16>& g, T (&E)[gtri(NBL)], T* cbuf2, T* ubuf, T* pwbuf, int gq, int npos)
{
    constexpr int GS = 16, MA = GS * NBL;
    const int k0 = GS * KB + 4 * gq;
    T* C = cbuf2 + (gq & 1) * 4 * MA;            // C[t * MA + i] = E(i, k0 + t)
    const bool bin = (g.b >> 2) == gq, ain = (g.a >> 2) == gq;      // this thread's column / row index (within a block) is one of the four
    const int tb = g.b & 3, sa = g.a & 3;
    // ---- A
    if (bin) {
#pragma unroll
        for (int li = KB; li < NBL; ++li) {
            const int i = GS * li + g.a;
            if (i > k0 + tb) C[tb * MA + i] = E[gidx(li, KB)];
        }
    }
    if (ain) {
#pragma unroll
        for (int lj = 0; lj <= KB; ++lj) {
            const int j = GS * lj + g.b;
            if (j < k0 + sa) C[sa * MA + j] = E[gidx(KB, lj)];
        }
        if (bin && sa == tb) C[sa * MA + k0 + sa] = E[gidx(KB, KB)];
    }
    GridPos<GS>::sync(blk);
    // ---- B  One entry of P per lane -- lane 4 s + t of every row of 16 lanes holds P[s][t]; a thread-private copy of
    // the block would be thirty more registers beside the 91 matrix entries of the largest instantiation -- swept by
    // lane exchanges, the same arithmetic in every row of lanes and every wave (so the failure decision is uniform);
    // the result, -P^-1, goes through an LDS copy of the wave's own.
    const int ln = blk.lane(), ps = (ln >> 2) & 3, pt = ln & 3;
    T pe = C[pt * MA + k0 + ps];
    int fail = 0;
    static_for<4>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        // row t of the block through scalar registers (four readlanes), column t through a DPP quad broadcast: no LDS
        // crossbar in the chain of the four pivots
        const T r0 = blk.bcast(pe, 4 * t), r1 = blk.bcast(pe, 4 * t + 1), r2 = blk.bcast(pe, 4 * t + 2), r3 = blk.bcast(pe, 4 * t + 3);
        const T d = t == 0 ? r0 : (t == 1 ? r1 : (t == 2 ? r2 : r3));
        const bool okp = (k0 + t < npos) ? (d > T(0)) : (d < T(0));
        if (fail == 0 && (!okp || !finite_(d))) fail = (k0 + t < npos) ? QPX_ST_Q_NOT_SPD : QPX_ST_A_RANK;
        const T r = rcp_(d);
        const T pit = blk.template quad_bcast<t>(pe);                       // P[s][t]
        const T ptj = pt == 0 ? r0 : (pt == 1 ? r1 : (pt == 2 ? r2 : r3));  // P[t][j]
        const T lm = pit * r;
        pe = (ps == t) ? ((pt == t) ? -r : ptj * r) : ((pt == t) ? lm : fma_(-lm, ptj, pe));
    });
    if (fail) return fail;
    const T pm = blk.shfl_xor(pe, (4 * ps + pt) ^ (4 * pt + ps));
    pe = ps >= pt ? pe : pm;                                 // symmetric to the bit: the lower triangle's values
    T* Pw = pwbuf + 16 * blk.uniform(blk.wave());
    if (ln < 16) Pw[ln] = pe;
    blk.wave_sync();
    const T pv = Pw[4 * sa + tb];                            // this thread's entry of -P^-1, if it holds one
    if (g.tid < MA) {
        T cs[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) cs[s] = C[s * MA + g.tid];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            T u = T(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) u = fma_(-cs[s], Pw[4 * s + t], u);      // Pw holds -P^-1
            ubuf[t * MA + g.tid] = u;
        }
    }
    GridPos<GS>::sync(blk);
    // ---- C
    const T ma = ain ? T(0) : T(1), mb = bin ? T(0) : T(1);               // the block's own rows / columns take no update
#pragma unroll 1                      // (unrolled, the four updates' operands are all fetched up front: 1.48 ms instead of 0.158 at C2, spills)
    for (int t = 0; t < 4; ++t) {
        const T* Ut = ubuf + t * MA;
        const T* Ct = C + t * MA;
        T vr[NBL];
#pragma unroll
        for (int l = 0; l < NBL; ++l) vr[l] = Ut[GS * l + g.a] * (l == KB ? ma : T(1));
        // (one column operand ahead, no further: left to itself the scheduler fetches all NBL of them up front, and the
        // largest instantiation has no registers for that beside its 91 matrix entries)
        T y = Ct[g.b] * (0 == KB ? mb : T(1));
#pragma unroll
        for (int lj = 0; lj < NBL; ++lj) {
            T yn = T(0);
            if (lj + 1 < NBL) yn = Ct[GS * (lj + 1) + g.b] * (lj + 1 == KB ? mb : T(1));
#pragma unroll
            for (int li = lj; li < NBL; ++li) E[gidx(li, lj)] = fma_(-vr[li], y, E[gidx(li, lj)]);
            QPX_SCHED_FENCE();
            y = yn;
        }
    }
    // (the addresses of the loads below are made opaque HERE: hoisted above the four updates, the 2 x NBL values they
    // fetch sat in registers through them -- 107 spilled registers at NBL = 13)
    int ua = g.a, ub = g.b;
    QPX_LAUNDER_V(ua);
    QPX_LAUNDER_V(ub);
    if (bin) {                                   // columns k0 + tb: (i, k0 + tb) = U_i,tb; inside the block: -P^-1
#pragma unroll
        for (int li = KB; li < NBL; ++li) {
            const T u = ubuf[tb * MA + GS * li + ua];
            E[gidx(li, KB)] = (li == KB && ain) ? pv : u;
        }
    }
    if (ain) {                                   // rows k0 + sa, columns outside the block: (k0 + sa, j) = U_j,sa
        // (one store per element, the block's own entry kept BY VALUE: the same store under two branches is merged
        // into one store through a pointer, and a register array whose address is taken lives in scratch memory)
#pragma unroll
        for (int lj = 0; lj <= KB; ++lj) {
            const T u = ubuf[sa * MA + GS * lj + ub];
            E[gidx(KB, lj)] = (lj == KB && bin) ? E[gidx(KB, lj)] : u;
        }
    }
    return 0;
}

template <class T, int NBL, int KB> struct SweepBlocks {
    static QPX_DEV int run(const Block& blk, const GridPos<16>& g, T (&E)[gtri(NBL)], T* vec2, T* dsl, T* cbuf2, T* ubuf, int npos, int npiv)
    {
        if (16 * KB >= npiv) return 0;
        constexpr int GSMA = 16 * NBL;
        const int kend = (npiv - 16 * KB < 16) ? (npiv - 16 * KB) : 16;
        int ka = 0;
#ifndef QPX_AB_RANK1_SWEEP
        // Four pivots per barrier pair where a CU holds at most two or three workgroups (NBL >= 10: <= 256 registers
        // per thread at 45 .. 91 matrix entries): there the barrier and the publish -> reciprocal chain of every pivot
        // are exposed.  Measured on one MI355X, pre-factorisation alone, rank-1 -> groups of four: C2 (NBL 13) 0.166 ->
        // 0.158 ms, C3 (NBL 10) 0.139 -> 0.123 ms; at NBL 8 with eight workgroups per CU (B = 2048, n = m = 64) the
        // other workgroups already hide those latencies and the groups' extra work (U, the assignments) costs 0.189 ->
        // 0.196 ms: rank-1 stays there (profiles/archive/r03w).
        if constexpr (NBL >= 10) {
#pragma unroll 1
            for (int gq = 0; 4 * gq + 4 <= kend; ++gq) {
                const int f = sweep_group4<T, NBL, KB>(blk, g, E, cbuf2, ubuf, ubuf + 4 * GSMA, gq, npos);
                if (f) return f;
                ka = 4 * gq + 4;
            }
        }
#endif
#pragma unroll 1
        for (; ka < kend; ++ka) {                // (the last n + q mod 4 pivots: one at a time)
            const int f = sweep_step<T, NBL, KB>(blk, g, E, vec2, dsl, ka, npos);
            if (f) return f;
        }
        return SweepBlocks<T, NBL, KB + 1>::run(blk, g, E, vec2, dsl, cbuf2, ubuf, npos, npiv);
    }
};
template <class T, int NBL> struct SweepBlocks<T, NBL, NBL> {
    static QPX_DEV int run(const Block&, const GridPos<16>&, T (&)[gtri(NBL)], T*, T*, T*, T*, int, int) { return 0; }
};

// ------------------------------------------------------------------------------------------
// Pre-factorisation by the symmetric SWEEP operator (replaces pre_factor_kkt, batch.py:375-429):
// the augmented matrix  S = [[Q, A^T, G^T], [A, 0, 0], [G, 0, 0]]  (order n+q+m) is held by a
// 16x16 thread grid in registers; sweeping the n pivots of Q and then the q pivots of the
// -A Q^-1 A^T block turns it, in place and by rank-1 updates only, into
//      [[ -K,   .,     . ],
//       [ -N^T, S11^-1, . ],
//       [  M,   W,    -R ]]      K = Q^-1 - Q^-1 A^T S11^-1 A Q^-1,  N = Q^-1 A^T S11^-1,
//                                M = G K,  W = G N,  R = G K G^T  (the reference's R)
// i.e. everything forward and backward need, with no triangular solve anywhere.
template <class T, int NBL>
QPX_DEV void sweep_body(const Block& blk, const PrefactorArgs<T>& a, int qp, T* lds)
{
    constexpr int GS = 16, NT = 256, MA = GS * NBL;
    const GridPos<GS> g(blk);
    const int n = a.n, m = a.m, q = a.q, nq = n + q, na = n + q + m;
    const FacLayout lay = fac_layout(n, m, q, a.images);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const In<T> Qg(a.Q, (size_t)qp * a.sQ, a.io32), Gg(a.G, (size_t)qp * a.sG, a.io32);
    const In<T> Ag(q > 0 ? a.A : nullptr, (size_t)qp * a.sA, a.io32);
    T* vec2 = lds;                 // 2*MA
    T* dsl = vec2 + 2 * MA;        // 8
    T* vout = dsl + 8;             // MA
    T* red = vout + MA;            // NBL*256

    QPX_PROF_INIT
    T E[gtri(NBL)];
    // ---- load the lower triangle of S (diagonal blocks in full)
#pragma unroll
    for (int li = 0; li < NBL; ++li)
#pragma unroll
        for (int lj = 0; lj <= li; ++lj) {
            int i = GS * li + g.a, j = GS * lj + g.b;
            if (j > i) { const int t = i; i = j; j = t; }       // upper part of a diagonal block: mirror
            T val = T(0);
            if (i < na && j < n) {
                if (i < n) val = Qg[(size_t)j * n + i];      // Q by its upper triangle: the coalesced read (lanes run over i)
                else if (i < nq) val = Ag[(size_t)(i - n) * n + j];
                else val = Gg[(size_t)(i - nq) * n + j];
            }
            E[gidx(li, lj)] = val;
        }
    QPX_PROF(0)
    // ---- || G^T 1 ||  (column sums of the G block, before it is swept)
    {
        T part[NBL];
#pragma unroll
        for (int lj = 0; lj < NBL; ++lj) {
            part[lj] = T(0);
#pragma unroll
            for (int li = lj; li < NBL; ++li) {
                const int i = GS * li + g.a, j = GS * lj + g.b;
                if (i >= nq && i < na && j < n) part[lj] += E[gidx(li, lj)];
            }
        }
        grid_reduce<T, GS, NBL, false, false>(blk, g, part, red, vout);
        if (blk.wave() == 0) {
            T acc = 0;
            for (int j = blk.lane(); j < n; j += kWave) acc = fma_(vout[j], vout[j], acc);
            acc = wave_sum(blk, acc);
            if (blk.lane() == 0) F[lay.scal] = sqrt_(acc);
        }
    }
    QPX_PROF(1)
    // ---- sweep pivots 0 .. n+q-1 (static block index via template recursion, see SweepBlocks)
    // (the four-pivot groups' buffers -- two generations of four columns, one of U, a 4 x 4 block per wave:
    // 12 MA + 64 <= 256 NBL -- share `red`, which only the column sums above used)
    const int fail = SweepBlocks<T, NBL, 0>::run(blk, g, E, vec2, dsl, red, red + 8 * MA, n, nq);
    GridPos<GS>::sync(blk);
    if (fail) {
        for (size_t e = blk.tid; e < lay.total; e += NT) F[e] = T(0);
        if (blk.tid == 0) a.status[qp] = fail;
        return;
    }
    QPX_PROF(2)
    // ---- scatter the blocks of the swept matrix to the blob
    const bool wRg = (a.images & 1) != 0, wRw = (a.images & 2) != 0 && lay.nbw > 0, wRm = (a.images & 4) != 0 && lay.nbt > 0;
    if (wRg)
        for (size_t e = blk.tid; e < grid_elems(16, lay.nbg); e += NT) F[lay.Rg + e] = T(0);
    if (wRw)
        for (size_t e = blk.tid; e < grid_elems(8, lay.nbw); e += NT) F[lay.Rw + e] = T(0);
    if (wRm)
        for (size_t e = blk.tid; e < tile_image_elems(lay.nbt); e += NT) F[lay.Rm + e] = T(0);
    GridPos<GS>::sync(blk);
#pragma unroll
    for (int li = 0; li < NBL; ++li)
#pragma unroll
        for (int lj = 0; lj <= li; ++lj) {
            const int i = GS * li + g.a, j = GS * lj + g.b;
            if (i >= na || j > i) continue;                        // lower triangle only (mirrors are written explicitly)
            const T val = E[gidx(li, lj)];
            if (i < n) {                                           // -K (n x n), both triangles
                F[lay.Kneg + (size_t)i * n + j] = val;
                F[lay.Kneg + (size_t)j * n + i] = val;
            } else if (i < nq) {
                if (j < n) F[lay.NTn + (size_t)(i - n) * n + j] = val;           // -N^T (q x n)
                else {                                                           // S11^-1 (q x q)
                    F[lay.S11i + (size_t)(i - n) * q + (j - n)] = val;
                    F[lay.S11i + (size_t)(j - n) * q + (i - n)] = val;
                }
            } else {
                const int zi = i - nq;
                if (j < n) {                                                     // M^T (n x m): the coalesced direction
                    F[lay.MT + (size_t)j * m + zi] = val;
                } else if (j < nq) {
                    F[lay.W + (size_t)zi * q + (j - n)] = val;                   // W (m x q)
                } else {                                                         // R = -block, grid layout of order m
                    const int zj = j - nq;
                    const int l2i = zi >> 4, ai = zi & 15, l2j = zj >> 4, bj = zj & 15;
                    if (wRg) {
                        T* blkp = F + lay.Rg + (size_t)(l2i * (l2i + 1) / 2 + l2j) * 256;
                        blkp[ai + 16 * bj] = -val;
                        if (l2i == l2j && zi != zj) blkp[bj + 16 * ai] = -val;
                    }
                    if (wRw) {                                                   // and of the 8x8 grid
                        const int w2i = zi >> 3, wa = zi & 7, w2j = zj >> 3, wb = zj & 7;
                        T* wp = F + lay.Rw + (size_t)(w2i * (w2i + 1) / 2 + w2j) * 64;
                        wp[wa + 8 * wb] = -val;
                        if (w2i == w2j && zi != zj) wp[wb + 8 * wa] = -val;
                    }
                    if (wRm) {                                                   // and of the matrix-core tiles
                        F[lay.Rm + tile_image_index(zi, zj)] = -val;
                        if (l2i == l2j && zi != zj) F[lay.Rm + tile_image_index(zj, zi)] = -val;
                    }
                }
            }
        }
    QPX_PROF(3)
    {
        const Block& b = blk;
        QPX_PROF_DUMP(F + lay.prof, T)
    }
    if (blk.tid == 0) a.status[qp] = 0;
}

QPX_LAYOUT_HD size_t lds_elems_sweep(int nbl) { return (size_t)3 * 16 * nbl + 8 + (size_t)nbl * 256; }

// ------------------------------------------------------------------------------------------
// The O(m) vector blocks of the PDIPM loop, by ONE wave on LDS-resident vectors (a vector of length m is NS slots of 64
// lanes); shared by the two orders of a pass in ipm_loop_role.
// Start point (batch.py:61-87): vX = -T^-1 c -> z, s shifted to >= 1 where their minimum is negative; z' = x
template <class T, int NS>
QPX_DEV void ipm_start_point(const Block& b, int lane, int m, int M8, const T* vX, T* vZ, T* vS, T* vA, T* vBZ, T* vBS, T* pSigz, T* pSigs)
{
    T x[NS];
    ld_slots<NS>(b, x, vX, m, T(0));
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
    const T sigz = (mnz < T(0)) ? (T(1) - mnz) : T(0);
    const T sigs = (mns < T(0)) ? (T(1) - mns) : T(0);
    if (lane == 0) { *pSigz = sigz; *pSigs = sigs; }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int i = k * kWave + lane;
        if (i < m) {
            const T zk = x[k] + sigz, sk = -x[k] + sigs;
            vZ[i] = zk; vS[i] = sk;
            vA[i] = x[k];
            vBZ[i] = zk; vBS[i] = sk;
        } else if (i < M8) {
            vA[i] = T(0);
        }
    }
}
// Affine step lengths, centring sigma, the corrector's right-hand side (batch.py:160-171)
template <class T, int NS>
QPX_DEV void ipm_affine_step(const Block& b, int lane, int m, T mu, T szdot, const T* vZ, const T* vS, const T* vRZ, const T* vRS,
                             const T* vD, const T* vX, T* vRSC, T* vRH, T* vDZA, T* vDSA)
{
    T z[NS], s[NS], rz[NS], rsv[NS], dd[NS], dza[NS], dsa[NS];
    ld_slots<NS>(b, z, vZ, m, T(1));
    ld_slots<NS>(b, s, vS, m, T(1));
    ld_slots<NS>(b, rz, vRZ, m, T(1));
    ld_slots<NS>(b, rsv, vRS, m, T(1));
    ld_slots<NS>(b, dd, vD, m, T(1));
    ld_slots<NS>(b, dza, vX, m, T(0));
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int i = k * kWave + lane;
        dsa[k] = (i < m) ? (-s[k] - dza[k] * dd[k]) : T(0);
    }
    T al = step_to_boundary_rcp<NS>(b, rz, dza, m);
    const T al2 = step_to_boundary_rcp<NS>(b, rsv, dsa, m);
    al = (al2 < al) ? al2 : al;
    al = (al < T(1)) ? al : T(1);
    T t3 = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int i = k * kWave + lane;
        if (i < m) t3 = fma_(s[k] + al * dsa[k], z[k] + al * dza[k], t3);
    }
    t3 = wave_sum(b, t3);
    T sig = t3 / szdot;
    sig = sig * sig * sig;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int i = k * kWave + lane;
        if (i < m) {
            const T rs = (-mu * sig + dsa[k] * dza[k]) * rsv[k];
            vRSC[i] = rs;
            vRH[i] = rs * dd[k];
            vDZA[i] = dza[k];
            vDSA[i] = dsa[k];
        }
    }
}
// The combined step with the 0.999 damping, tau, z' (batch.py:173-198)
template <class T, int NS>
QPX_DEV void ipm_final_step(const Block& b, int lane, int m, T* vZ, T* vS, const T* vRZ, const T* vRS, const T* vD, const T* vX,
                            const T* vRSC, const T* vDZA, const T* vDSA, T* vA, T* pTau, T* pAlphaPrev, T sigz)
{
    T z[NS], s[NS], rz[NS], rsv[NS], dz[NS], ds[NS];
    ld_slots<NS>(b, z, vZ, m, T(1));
    ld_slots<NS>(b, s, vS, m, T(1));
    ld_slots<NS>(b, rz, vRZ, m, T(1));
    ld_slots<NS>(b, rsv, vRS, m, T(1));
    ld_slots<NS>(b, dz, vX, m, T(0));
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int i = k * kWave + lane;
        const T dsc = (i < m) ? ((-vRSC[i] - dz[k]) * vD[i]) : T(0);
        dz[k] = (i < m) ? (vDZA[i] + dz[k]) : T(0);
        ds[k] = (i < m) ? (vDSA[i] + dsc) : T(0);
    }
    T al = step_to_boundary_rcp<NS>(b, rz, dz, m);
    const T al3 = step_to_boundary_rcp<NS>(b, rsv, ds, m);
    al = (al3 < al) ? al3 : al;
    al = T(0.999) * al;
    al = (al < T(1)) ? al : T(1);
    const T tau = (T(1) - al) * *pTau;
    const T tsz = tau * sigz;
    b.wave_sync();           // every lane has read the old tau
    if (lane == 0) { *pTau = tau; *pAlphaPrev = al; }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int i = k * kWave + lane;
        if (i < m) {
            const T zk = fma_(al, dz[k], z[k]);
            vZ[i] = zk;
            vS[i] = fma_(al, ds[k], s[k]);
            vA[i] = zk - tsz;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The PDIPM loop on the format-3 blob.  Mathematics and control flow: see ipm_body (same
// reference citations); the factorisation is ldl_inv and every solve is two triangular mat-vecs.
template <class T, class Mat, int NS, class P>
QPX_DEV void ipm_loop_role(const Block& b, const IpmArgs<T>& a, int qp, T* lds, const P& g)
{
    constexpr int M8 = Mat::MP, NT = Mat::NT;
    const int n = a.n, m = a.m, q = a.q;
    const FacLayout lay = fac_layout(n, m, q, a.images);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const T* Rg = Mat::image(F, lay);
    // compile-time vector stride (64 NS >= max(n, MP, q)): every LDS vector is then a constant offset
    // from one base and the compiler addresses them all with one lane register + immediates (with a
    // run-time stride it hoisted ~40 per-vector address registers out of the loop and spilled them)
    constexpr size_t v = 64 * (size_t)NS;
    T* rd = lds;          // 1/d_k (M8)
    T* vA = rd + v;       // z' (M8)
    T* vB = vA + v;       // R z' (M8)
    T* vC = vB + v;       // c (M8, zero padded)
    T* vR1 = vC + v;      // R 1 (M8)
    T* vD = vR1 + v;      // s/z, 1 on the pad (M8)
    T* vBZ = vD + v;      // best z
    T* vBS = vBZ + v;     // best s
    // (r6) seventeen vectors, not twenty: three share LDS with one whose life does not overlap theirs -- at one wave per
    // QP and four tile rows those 1.5 KB decide between six and eight workgroups on a CU (qpx_tile.h: kInPlace)
    T* vRH = vB;          // right-hand side of a solve (M8, zero padded): formed FROM R z' element by element, in place
    T* vX = vBS + v;      // its solution
    T* vTm = vX + v;      // scratch of the solve
    T* vP = vX;           // p staging (n): read by the two products in front of the loop only (vX is first written behind them); the epilogue stages p again, in vRZ
    T* vZ = vTm + v;      // z, s and their reciprocals
    T* vS = vZ + v;
    T* vRZ = vS + v;
    T* vRS = vRZ + v;
    T* vDZA = vA;         // affine step: written behind the mat-vec (the last reader of z' in a pass), read by the block that then writes the new z'
    T* vDSA = vRS + v;    // ... and the corrector right-hand side rs
    T* vRSC = vDSA + v;
    T* vX0 = vRSC + v;    // x0 = -K p (n): formed with c at the start, while p's products share their loads' flight
    T* sc = vX0 + v;                                // 16 scalars of the IPM state
    int* ctrl = reinterpret_cast<int*>(sc + 16);    // 8 elements of control words
    T* scr = sc + 24;     // Mat::scratch_elems()

    const int lane = b.lane();
    const bool w0 = g.lead(b);        // the wave that does the O(m) vector work
    const T mT = (T)m;
    const int io32 = a.io32;
    const In<T> pg(a.p, (size_t)qp * a.sp, io32), hg(a.h, (size_t)qp * a.sh, io32);
    const In<T> bg(q > 0 ? a.b : nullptr, (size_t)qp * a.sb, io32);

    if (a.status[qp] & (QPX_ST_Q_NOT_SPD | QPX_ST_A_RANK)) {
        const T nanv = Lim<T>::inf() - Lim<T>::inf();
        for (int i = b.tid; i < n; i += b.nt) put_(a.zhat, io32, (size_t)qp * n + i, nanv);
        for (int i = b.tid; i < m; i += b.nt) {
            put_(a.lam, io32, (size_t)qp * m + i, nanv);
            put_(a.slack, io32, (size_t)qp * m + i, nanv);
        }
        for (int i = b.tid; i < q; i += b.nt) put_(a.nu, io32, (size_t)qp * q + i, nanv);
        if (b.tid == 0) {
            a.iters[qp] = 0;
            put_(a.best_resid, io32, (size_t)qp, Lim<T>::inf());
        }
        return;
    }
    QPX_PROF_INIT
    // ---- c = h - G x0 = h + M p - W b   (x0 = -K p + N b is formed only at the end)
    for (int i = b.tid; i < n; i += NT) vP[i] = pg[i];
    for (int i = b.tid; i < M8; i += NT) {
        vC[i] = (i < m) ? hg[i] : T(0);
        vD[i] = T(1);
        vA[i] = T(1);                     // first use: R 1
    }
    Mat::sync(b);
    block_matTvec2<T, 1, 0>(b, vC, F + lay.MT, m, vX0, F + lay.Kneg, n, vP, n);      // c += M p;  x0 = -K p
    if (q > 0) {
        Mat::sync(b);
        for (int i = b.tid; i < q; i += NT) vTm[i] = bg[i];
        Mat::sync(b);
        for (int j = b.tid; j < m; j += NT) {
            T acc = vC[j];
            for (int r = 0; r < q; ++r) acc = fma_(-F[lay.W + (size_t)j * q + r], vTm[r], acc);
            vC[j] = acc;
        }
    }
    Mat::sync(b);
    QPX_PROF(0)

    // The IPM state of wave 0 (z, s, their reciprocals, step directions, scalars) lives in LDS between
    // the blocks that use it, not in registers: it is touched for a few hundred instructions per
    // iteration, and carried through the factorisation it cost 32 spilled VGPRs (scratch reloads with
    // a memory round trip each) -- measured on MI355X.
    typename Mat::Regs E;
    enum { kTau = 0, kBtau, kSigz, kSigs, kBres, kFeasPrev, kAlphaPrev, kMu, kSzdot, kFeas, kResid, kGt1 };
    enum { kStop = 0, kNnot, kFloor, kSt, kIters };
    if (b.tid == 0) {
        sc[kTau] = T(1); sc[kBtau] = T(1); sc[kSigz] = T(0); sc[kSigs] = T(0); sc[kBres] = Lim<T>::inf();
        sc[kFeasPrev] = T(0); sc[kAlphaPrev] = T(0);
        sc[kGt1] = F[lay.scal];          // || G^T 1 ||: read from the blob once, not once per iteration (a global load on the chain)
        ctrl[kStop] = 0; ctrl[kNnot] = 0; ctrl[kFloor] = 0; ctrl[kSt] = 0; ctrl[kIters] = 0;
    }
    for (int i = b.tid; i < M8; i += NT) {
        vZ[i] = T(1); vS[i] = T(1); vRZ[i] = T(1); vRS[i] = T(1);
        vDSA[i] = T(0); vRSC[i] = T(0);
    }
    Mat::sync(b);

    // ---- pass -1 is the start point: T = R + I, z_i = -T^-1 c, s_i = -z_i, shifts (batch.py:61-87),
    // and R 1 on the way; passes 0.. are the IPM iterations.  One loop so that the factorisation and
    // the solves are instantiated once (they are the bulk of the kernel's code).
    // (Re-loading R right after the last solve of the previous pass, so that the loads fly during the lead wave's
    // vector work, was measured twice: no gain in round 2 -- profiles/archive/r02b_panel_ab.txt, "late" -- and +3 % loop time
    // with the chain-wave form, profiles/archive/r03h_ab_loop_variants.txt.)
    int stop = 0;
    if constexpr (Mat::kAhead) {
    // ---- (r6) the same passes with the chain wave AHEAD of the tile waves (TileMat::kAhead): d = s/z of a pass is final when
    // the previous pass ends, so the chain wave eliminates pivot block 0 while the tile waves load R and form R z', and the
    // residual / best-iterate / stop bookkeeping runs on it in panel 0's second interval, beside the tile waves' longest
    // stretch of updates.  The stop decision is read by every wave behind that panel's last barrier; the panel the waves
    // ran ahead of it on a pass that stops is dropped (T lives in registers nobody reads again).  Mathematics and the order of
    // the floating-point operations as written: those of the order below.
    typename Mat::Ahead ah;
    Mat::ahead_init(b, g, ah, Rg);
    for (int it = -1; it < a.maxIter && !stop; ++it) {
        const bool first = it < 0;
        Mat::load(b, g, E, Rg);
        QPX_PROF(2)
        if (w0) {
            if (!first) {
                // 1/z, 1/s, d = s/z and mu of the iterate the previous pass left (every later division by z or s is a multiplication)
                T szdot = 0;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int i = k * kWave + lane;
                    if (i < m) {
                        const T zk = vZ[i], sk = vS[i];
                        szdot = fma_(sk, zk, szdot);
                        const T rzk = rcp_(zk);
                        vRZ[i] = rzk;
                        vRS[i] = rcp_(sk);
                        vD[i] = sk * rzk;
                    }
                }
                szdot = wave_sum(b, szdot);
                if (lane == 0) { sc[kMu] = abs_(szdot / mT); sc[kSzdot] = szdot; }
                b.wave_sync();
            }
            Mat::ahead_pivot0(b, g, ah, vD, scr, rd, m);
        }
        QPX_PROF(4)
        Mat::ahead_front(b, g, E, vA, vZ, vS, scr);      // tile waves: partial sums of R z' (first pass: R 1), T = R + diag(s/z), panel 0's old rows
        Mat::sync(b);
        QPX_PROF(3)                  // (profiling build, chain wave: its wait for the tile waves)
        const int rc = Mat::ldl_inv_ahead(b, g, E, scr, rd, m, [&] { Mat::ahead_gather(b, lane, scr, first ? vR1 : vB); }, [&] {
            if (first) return;
            const T tsz = sc[kTau] * sc[kSigz];
            T pri2 = 0;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                if (i < m) {
                    const T rz = vS[i] - vC[i] - vB[i];
                    pri2 = fma_(rz, rz, pri2);
                }
            }
            pri2 = wave_sum(b, pri2);
            const T mu = sc[kMu];
            const T pri = sqrt_(pri2);
            const T dual = tsz * sc[kGt1];         // || G^T 1 ||
            const T feas = pri + dual, resid = feas + mT * mu;
            // best iterate and the stop decision (batch.py:118-143)
            const T tau = sc[kTau];
            T bres = sc[kBres];
            int nnot = ctrl[kNnot], floor_hit = ctrl[kFloor];
            int stopf = 0;
            const bool better = (it == 0) || (resid < bres);
            if (better) {
                bres = resid; nnot = 0;
                for (int i = lane; i < m; i += kWave) { vBZ[i] = vZ[i]; vBS[i] = vS[i]; }
            } else if (a.stall_policy == 1 || (a.stall_policy == 2 && mT * mu < feas)) {
                nnot += 1;
            } else {
                nnot = 0;
            }
            if (a.stall_policy == 2 && it >= 1 && feas > T(2) * (T(1) - sc[kAlphaPrev]) * sc[kFeasPrev]) floor_hit = 1;
            if ((a.stall_policy != 0 && nnot >= a.notImprovedLim) || bres < a.eps || mu > T(1e32)) stopf = 1;
            if (a.stall_policy == 2 && floor_hit && mT * mu < T(1e-2) * feas) stopf = 1;
            const bool bad = !finite_(resid);
            if (bad) stopf = 1;
            b.wave_sync();           // every lane has read the scalars lane 0 is about to replace
            if (lane == 0) {
                ctrl[kIters] = it + 1;
                if (better) { sc[kBres] = bres; sc[kBtau] = tau; }
                sc[kFeasPrev] = feas;
                ctrl[kNnot] = nnot; ctrl[kFloor] = floor_hit;
                if (bad) ctrl[kSt] |= QPX_ST_NONFINITE;
                ctrl[kStop] = stopf;
                if (a.trace) {
                    const size_t tr = ((size_t)it * a.B + qp) * 3;
                    put_(a.trace, io32, tr, pri); put_(a.trace, io32, tr + 1, dual); put_(a.trace, io32, tr + 2, mu);
                }
            }
        }, [&] {
            if (first) return;
            const T tsz = sc[kTau] * sc[kSigz];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                if (i < m) vRH[i] = vC[i] + vB[i] + tsz * vR1[i];       // affine right-hand side c + R z
            }
        }, [&] { return first ? 0 : ctrl[kStop]; });
        QPX_PROF(5)
        if (rc < 0) { stop = 1; break; }
        if (rc > 0) {                // uniform: every wave reads the same flag
            if (w0) {
                if (lane == 0) ctrl[kSt] |= QPX_ST_KKT_BREAKDOWN;
                if (first) {
                    for (int i = lane; i < M8; i += kWave) {
                        if (i < m) { vBZ[i] = T(1); vBS[i] = T(1); }
                        vA[i] = T(0);
                    }
                }
            }
            break;
        }
        QPX_PROF(1)
        Mat::template solve_neg<true>(b, g, E, rd, m, first ? vC : vRH, vX, vTm, scr);
        QPX_PROF(6)
        if (first) {
            if (w0) ipm_start_point<T, NS>(b, lane, m, M8, vX, vZ, vS, vA, vBZ, vBS, sc + kSigz, sc + kSigs);
            Mat::sync(b);
            QPX_PROF(1)
            continue;
        }
        if (w0) ipm_affine_step<T, NS>(b, lane, m, sc[kMu], sc[kSzdot], vZ, vS, vRZ, vRS, vD, vX, vRSC, vRH, vDZA, vDSA);
        Mat::sync(b);
        QPX_PROF(1)
        Mat::template solve_neg<true>(b, g, E, rd, m, vRH, vX, vTm, scr);
        QPX_PROF(6)
        // (The next pass's copy of R requested HERE by the tile waves -- their registers are free once the second solve has
        // read them -- so that it arrives under the chain wave's step-length block: measured again in round 6 with the chain
        // wave ahead, 0.4448 against 0.4428 ms, no gain -- profiles/r06g_ab_early_load_two_accumulators.txt; rounds 2 and 3
        // had measured the same with their orders of a pass.)
        if (w0) ipm_final_step<T, NS>(b, lane, m, vZ, vS, vRZ, vRS, vD, vX, vRSC, vDZA, vDSA, vA, sc + kTau, sc + kAlphaPrev, sc[kSigz]);
        Mat::sync(b);
        QPX_PROF(1)
    }
    } else {
    for (int it = -1; it < a.maxIter && !stop; ++it) {
        const bool first = it < 0;
        Mat::load(b, g, E, Rg);
        QPX_PROF(2)
        Mat::template symv<true>(b, g, E, vA, first ? vR1 : vB, scr);       // first pass: vA = 1; read by the lead wave only
        QPX_PROF(3)
        if (w0 && !first) {
            const T tsz = sc[kTau] * sc[kSigz];
            T pri2 = 0, szdot = 0;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                if (i < m) {
                    const T zk = vZ[i], sk = vS[i];
                    const T rz = sk - vC[i] - vB[i];
                    pri2 = fma_(rz, rz, pri2);
                    szdot = fma_(sk, zk, szdot);
                    vRH[i] = vC[i] + vB[i] + tsz * vR1[i];       // affine right-hand side c + R z
                    // 1/z, 1/s and d = s/z once per iteration: every later division by z or s is a multiplication
                    const T rzk = rcp_(zk);
                    vRZ[i] = rzk;
                    vRS[i] = rcp_(sk);
                    vD[i] = sk * rzk;
                }
            }
            pri2 = wave_sum(b, pri2);
            szdot = wave_sum(b, szdot);
            const T mu = abs_(szdot / mT);
            const T pri = sqrt_(pri2);
            const T dual = tsz * sc[kGt1];         // || G^T 1 ||
            const T feas = pri + dual, resid = feas + mT * mu;
            // best iterate and the stop decision, BEFORE the factorisation (as batch.py:118-143 does): the
            // pass that stops costs a mat-vec, not a factorisation
            const T tau = sc[kTau];
            T bres = sc[kBres];
            int nnot = ctrl[kNnot], floor_hit = ctrl[kFloor];
            int stopf = 0;
            const bool better = (it == 0) || (resid < bres);
            if (better) {
                bres = resid; nnot = 0;
                for (int i = lane; i < m; i += kWave) { vBZ[i] = vZ[i]; vBS[i] = vS[i]; }
            } else if (a.stall_policy == 1 || (a.stall_policy == 2 && mT * mu < feas)) {
                nnot += 1;
            } else {
                nnot = 0;
            }
            if (a.stall_policy == 2 && it >= 1 && feas > T(2) * (T(1) - sc[kAlphaPrev]) * sc[kFeasPrev]) floor_hit = 1;
            if ((a.stall_policy != 0 && nnot >= a.notImprovedLim) || bres < a.eps || mu > T(1e32)) stopf = 1;
            if (a.stall_policy == 2 && floor_hit && mT * mu < T(1e-2) * feas) stopf = 1;
            const bool bad = !finite_(resid);
            if (bad) stopf = 1;
            b.wave_sync();           // every lane has read the scalars lane 0 is about to replace
            if (lane == 0) {
                sc[kMu] = mu; sc[kSzdot] = szdot;
                ctrl[kIters] = it + 1;
                if (better) { sc[kBres] = bres; sc[kBtau] = tau; }
                sc[kFeasPrev] = feas;
                ctrl[kNnot] = nnot; ctrl[kFloor] = floor_hit;
                if (bad) ctrl[kSt] |= QPX_ST_NONFINITE;
                ctrl[kStop] = stopf;
                if (a.trace) {
                    const size_t tr = ((size_t)it * a.B + qp) * 3;
                    put_(a.trace, io32, tr, pri); put_(a.trace, io32, tr + 1, dual); put_(a.trace, io32, tr + 2, mu);
                }
            }
        }
        Mat::sync(b);
        stop = first ? 0 : ctrl[kStop];
        if (stop) break;
        Mat::add_diag(g, E, vD);
        QPX_PROF(4)
        const bool ok = Mat::ldl_inv(b, g, E, scr, rd, m);
        QPX_PROF(5)
        if (!ok) {                   // uniform: every wave computes the same pivots
            if (w0) {
                if (lane == 0) ctrl[kSt] |= QPX_ST_KKT_BREAKDOWN;
                if (first) {
                    for (int i = lane; i < M8; i += kWave) {
                        if (i < m) { vBZ[i] = T(1); vBS[i] = T(1); }
                        vA[i] = T(0);
                    }
                }
            }
            break;
        }
        // first pass: z_i = -T^-1 c; iterations: affine scaling direction dz_aff = -T^-1 (c + R z)
        QPX_PROF(1)
        Mat::template solve_neg<true>(b, g, E, rd, m, first ? vC : vRH, vX, vTm, scr);
        QPX_PROF(6)
        if (first) {
            if (w0) ipm_start_point<T, NS>(b, lane, m, M8, vX, vZ, vS, vA, vBZ, vBS, sc + kSigz, sc + kSigs);
            Mat::sync(b);
            QPX_PROF(1)
            continue;
        }
        if (w0) ipm_affine_step<T, NS>(b, lane, m, sc[kMu], sc[kSzdot], vZ, vS, vRZ, vRS, vD, vX, vRSC, vRH, vDZA, vDSA);
        Mat::sync(b);
        QPX_PROF(1)
        Mat::template solve_neg<true>(b, g, E, rd, m, vRH, vX, vTm, scr);
        QPX_PROF(6)
        if (w0) ipm_final_step<T, NS>(b, lane, m, vZ, vS, vRZ, vRS, vD, vX, vRSC, vDZA, vDSA, vA, sc + kTau, sc + kAlphaPrev, sc[kSigz]);
        Mat::sync(b);
        QPX_PROF(1)
    }
    }

    // ---- outputs (batch.py:143,207)
    if (w0) {
        const T bres = sc[kBres];
        int st = ctrl[kSt];
        const int iters = ctrl[kIters];
        if (iters >= a.maxIter && !(bres < a.eps)) st |= QPX_ST_MAXITER;
        if (!(bres <= T(1))) st |= QPX_ST_INACCURATE;
        const T bts = sc[kBtau] * sc[kSigz];
        for (int i = lane; i < m; i += kWave) {
            const T bz = vBZ[i];
            put_(a.lam, io32, (size_t)qp * m + i, bz);
            put_(a.slack, io32, (size_t)qp * m + i, vBS[i]);
            vA[i] = bz - bts;
        }
        if (lane == 0) {
            a.iters[qp] = iters;
            a.status[qp] |= st;
            put_(a.best_resid, io32, (size_t)qp, bres);
        }
    }
    if (q > 0) {
        for (int i = b.tid; i < q; i += NT) vTm[i] = bg[i];
        for (int i = b.tid; i < n; i += NT) vRZ[i] = pg[i];      // p once more (nu needs it; its staging vector of the prologue is long gone): 1/z is not needed any more
    }
    Mat::sync(b);
    // zhat = x0 - M^T z' = -K p + N b - M^T z'
    for (int i = b.tid; i < n; i += NT) vX[i] = vX0[i];
    Mat::sync(b);
    block_matvec16<T, 2>(b, vX, F + lay.MT, vA, n, m);
    if (q > 0) {
        Mat::sync(b);
        block_matTvec<T, 2>(b, vX, F + lay.NTn, vTm, q, n);
    }
    Mat::sync(b);
    for (int i = b.tid; i < n; i += NT) put_(a.zhat, io32, (size_t)qp * n + i, vX[i]);
    if (q > 0) {
        // nu = -S11^-1 b + NTn p - W^T z'
        for (int r = b.tid; r < q; r += NT) {
            T acc = 0;
            for (int c2 = 0; c2 < q; ++c2) acc = fma_(-F[lay.S11i + (size_t)r * q + c2], vTm[c2], acc);
            for (int k = 0; k < n; ++k) acc = fma_(F[lay.NTn + (size_t)r * n + k], vRZ[k], acc);
            for (int j = 0; j < m; ++j) acc = fma_(-F[lay.W + (size_t)j * q + r], vA[j], acc);
            put_(a.nu, io32, (size_t)qp * q + r, acc);
        }
    }
    QPX_PROF(7)
    QPX_PROF_DUMP(a.trace ? a.trace + (size_t)qp * 8 : (T*)nullptr, T)
}

// The loop body runs once per wave role (the tile kernels' chain-wave form: Mat::with_role); every other form has one.
template <class T, class Mat, int NS>
QPX_DEV void ipm_loop_body(const Block& b, const IpmArgs<T>& a, int qp, T* lds)
{
    typename Mat::Pos g(b);
    g.assign(b, reinterpret_cast<int*>(lds));      // (an LDS word nothing else uses before the barriers inside)
    Mat::with_role(g, [&](const auto& gp) { ipm_loop_role<T, Mat, NS>(b, a, qp, lds, gp); });
}

template <class T, int GS, int NBL, int NS>
QPX_DEV void ipm_grid_body(const Block& b, const IpmArgs<T>& a, int qp, T* lds)
{
    ipm_loop_body<T, GridMat<T, GS, NBL>, NS>(b, a, qp, lds);
}

// ------------------------------------------------------------------------------------------
// factor_kkt + solve_kkt for arbitrary right-hand sides and QPFunctionFn.backward on the
// format-3 blob (see kkt_body for the reference citations):
//   dz = -T^-1 (M rx + W ry + rs/d - rz),  dx = -K rx - M^T dz - N ry,
//   dy = S11^-1 ry - N^T rx - W^T dz,      ds = (-rs - dz)/d
template <class T, class Mat, bool kBackward, class P>
QPX_DEV void kkt_mat_role(const Block& b, const KktArgs<T>& a, int qp, T* lds, const P& g)
{
    constexpr int M8 = Mat::MP, NT = Mat::NT;
    const int n = a.n, m = a.m, q = a.q;
    const FacLayout lay = fac_layout(n, m, q, a.images);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const size_t v = align4(max2(max2((size_t)n, (size_t)M8), (size_t)q));
    T* rd = lds;
    T* vD = rd + v;       // 1/d, 1 on the pad
    T* vRX = vD + v;      // rx (n)
    T* vRY = vRX + v;     // ry (q)
    T* vRH = vRY + v;     // right-hand side of the solve (M8)
    T* vDZ = vRH + v;     // dz (M8)
    T* vTm = vDZ + v;
    T* vDX = vTm + v;     // dx (n)
    T* vDY = vDX + v;     // dy (q)
    T* vZH = vDY + v;     // zhat (n)   [backward]
    T* vLM = vZH + v;     // lam (m)    [backward]
    T* vNU = vLM + v;     // nu (q)     [backward]
    T* vCX = vNU + v;     // corrections of the iterative refinement (n, M8, q)
    T* vCZ = vCX + v;
    T* vCY = vCZ + v;
    T* scr = vCY + v;     // Mat::scratch_elems()

    const int io32 = a.io32;
    const In<T> rxg(kBackward ? a.dl_dz : a.rx, (size_t)qp * n, io32);
    const In<T> rsg(kBackward ? nullptr : a.rs, (size_t)qp * m, io32);
    const In<T> rzg(kBackward ? nullptr : a.rz, (size_t)qp * m, io32);
    const In<T> ryg((!kBackward && q > 0) ? a.ry : nullptr, (size_t)qp * q, io32);
    const In<T> lamg(kBackward ? a.lam : nullptr, (size_t)qp * m, io32), slg(kBackward ? a.slack : nullptr, (size_t)qp * m, io32);
    const In<T> dg(kBackward ? nullptr : a.d, (size_t)qp * m, io32);
    QPX_PROF_INIT

    for (int i = b.tid; i < n; i += NT) vRX[i] = rxg ? rxg[i] : T(0);
    for (int i = b.tid; i < q; i += NT) vRY[i] = ryg ? ryg[i] : T(0);
    for (int i = b.tid; i < M8; i += NT) {
        T dinv = T(1), rhs = T(0);
        if (i < m) {
            T d;
            if (kBackward) {
                const T l = lamg[i], sl = slg[i];
                d = ((l < T(1e-8)) ? T(1e-8) : l) / ((sl < T(1e-8)) ? T(1e-8) : sl);      // qp.py:148
            } else {
                d = dg[i];
            }
            dinv = T(1) / d;
            rhs = (rsg ? rsg[i] * dinv : T(0)) - (rzg ? rzg[i] : T(0));
        }
        vD[i] = dinv;
        vRH[i] = rhs;
    }
    Mat::sync(b);
    // One application of the condensed KKT inverse: inputs rX (n), rY (q) and rH = rs/d - rz (M8),
    //   oZ = -T^-1 (rH + M rX + W rY),  oX = -K rX - M^T oZ - N rY,  oY = S11^-1 rY - N^T rX - W^T oZ   (rH is overwritten)
    // in two halves: the products with rX, which need no factor, and the rest.  (r4) Both products at once -- rH += M rX
    // and oX = -K rX, sixteen rows in flight per thread, all threads busy; one after the other with eight in flight they
    // were round trips to memory one behind the other, and -K rX waited for a solve it does not need -- and, the first
    // time, BEFORE the matrix is loaded: beside the tiles their 64 registers cost a workgroup per CU at four tile rows.
    auto products = [&](const T* rX, const T* rY, T* rH, T* oX) {
        // (eight in flight at up to four tile rows: sixteen made the kernel's register peak there, 142 -> 178, and cost
        // the third workgroup per CU -- +1.4 % on the step at B = 8192, nz = nineq = 64, profiles/archive/r04t)
        block_matTvec2<T, 1, 0, T, (M8 > 64 ? 16 : 8)>(b, rH, F + lay.MT, m, oX, F + lay.Kneg, n, rX, n);
        if (q > 0) {
            Mat::sync(b);
            for (int j = b.tid; j < m; j += NT) {
                T acc = rH[j];
                for (int r = 0; r < q; ++r) acc = fma_(F[lay.W + (size_t)j * q + r], rY[r], acc);
                rH[j] = acc;
            }
        }
        Mat::sync(b);
    };
    QPX_PROF(0)
    products(vRX, vRY, vRH, vDX);
    QPX_PROF(1)
    // (round 5 tried the image of R requested BEFORE the products at seven tile rows, where the registers allow it, so that its
    // round trip to HBM runs under theirs: 1.5 us SLOWER at C2, 0.0499 vs 0.0485 ms, same box -- the image's 57 KB then
    // compete with the 160 KB the products stream.  profiles/r05f_ab_symv_prefetch_and_backward_load_order.txt)
    typename Mat::Regs E;
    bool ok;
    if constexpr (Mat::kAhead) {
        // (r6) chain-wave form: the chain wave eliminates pivot block 0 -- R(0,0) from its own five loads + 1/d -- while the
        // tile waves fetch their 57 KB of R; one barrier in front of the panels instead of three (TileMat::kAhead)
        typename Mat::Ahead ah;
        Mat::ahead_init(b, g, ah, Mat::image(F, lay));
        Mat::load(b, g, E, Mat::image(F, lay));
        Mat::ahead_pivot0(b, g, ah, vD, scr, rd, m);
        Mat::ahead_publish0(b, g, E, vD, scr);
        Mat::sync(b);
        QPX_PROF(2)
        ok = Mat::ldl_inv_ahead(b, g, E, scr, rd, m, [] {}, [] {}, [] {}, [] { return 0; }) == 0;
    } else {
        Mat::load(b, g, E, Mat::image(F, lay));
        Mat::add_diag(g, E, vD);
        QPX_PROF(2)
        ok = Mat::ldl_inv(b, g, E, scr, rd, m);
    }
    QPX_PROF(3)
    if (!ok && b.tid == 0 && a.status) a.status[qp] |= QPX_ST_KKT_BREAKDOWN;
    auto finish = [&](const T* rX, const T* rY, T* rH, T* oZ, T* oX, T* oY) {
        if (ok) Mat::solve_neg(b, g, E, rd, m, rH, oZ, vTm, scr);
        else {
            for (int i = b.tid; i < M8; i += NT) oZ[i] = T(0);
            Mat::sync(b);
        }
        QPX_PROF(4)
        // oX = Kneg rX - M^T oZ + NTn^T rY     (Kneg = -K, NTn = -N^T; the first term from above)
        block_matvec16<T, 2>(b, oX, F + lay.MT, oZ, n, m);
        if (q > 0) {
            Mat::sync(b);
            block_matTvec<T, 1>(b, oX, F + lay.NTn, rY, q, n);
            for (int r = b.tid; r < q; r += NT) {
                T acc = 0;
                for (int c2 = 0; c2 < q; ++c2) acc = fma_(F[lay.S11i + (size_t)r * q + c2], rY[c2], acc);
                for (int k = 0; k < n; ++k) acc = fma_(F[lay.NTn + (size_t)r * n + k], rX[k], acc);
                for (int j = 0; j < m; ++j) acc = fma_(-F[lay.W + (size_t)j * q + r], oZ[j], acc);
                oY[r] = acc;
            }
        }
        Mat::sync(b);
    };
    auto apply = [&](const T* rX, const T* rY, T* rH, T* oZ, T* oX, T* oY) {
        products(rX, rY, rH, oX);
        finish(rX, rY, rH, oZ, oX, oY);
    };
    finish(vRX, vRY, vRH, vDZ, vDX, vDY);
    QPX_PROF(5)

    // Iterative refinement on the residual of the ORIGINAL KKT system (batch.py:244-270, solve_kkt_ir; the factor
    // is re-used, not re-computed as there):  res = K sol + rhs  with the caller's Q, G, A,  sol += K~^-1 (-res).
    //   resx = Q dx + G^T dz + A^T dy + rx,   resz = G dx + ds + rz  (ds = (-rs - dz)/d, so ress = 0),   resy = A dx + ry
    if (a.refine > 0 && a.Q && a.G && !io32) {
        const T* Qg = a.Q + (size_t)qp * a.sQ;
        const T* Gg = a.G + (size_t)qp * a.sG;
        const T* Ag = (q > 0 && a.A) ? a.A + (size_t)qp * a.sA : nullptr;
        for (int it = 0; it < a.refine; ++it) {
            for (int i = b.tid; i < n; i += NT) vRX[i] = rxg ? rxg[i] : T(0);
            for (int i = b.tid; i < q; i += NT) vRY[i] = ryg ? ryg[i] : T(0);
            for (int i = b.tid; i < M8; i += NT)     // -(ds + rz): the part of -resz that needs no matrix
                vRH[i] = (i < m) ? -((-(rsg ? rsg[i] : T(0)) - vDZ[i]) * vD[i] + (rzg ? rzg[i] : T(0))) : T(0);
            Mat::sync(b);
            // residuals accumulate in double whatever T is: fixed-precision refinement cannot improve the forward
            // error of an ill-conditioned solve (cond(Q) ~ 1e6 on the benchmark generator), mixed precision can
            block_matTvec<T, 1, double>(b, vRX, Qg, vDX, n, n);           // Q symmetric: column-parallel over its rows
            Mat::sync(b);
            block_matTvec<T, 1, double>(b, vRX, Gg, vDZ, m, n);           // + G^T dz
            block_matvec<T, 2, double>(b, vRH, Gg, vDX, m, n);            // - G dx
            if (Ag) {
                Mat::sync(b);
                block_matTvec<T, 1, double>(b, vRX, Ag, vDY, q, n);       // + A^T dy
                block_matvec<T, 1, double>(b, vRY, Ag, vDX, q, n);        // + A dx
            }
            Mat::sync(b);
            // correction = K~^-1 (-res):  apply() solves K c = -(rX, ., rH-part, rY) for right-hand sides given as
            // (rx, rs/d - rz, ry), so pass resx, -resz (rs = 0), resy
            apply(vRX, vRY, vRH, vCZ, vCX, vCY);
            for (int i = b.tid; i < n; i += NT) vDX[i] += vCX[i];
            for (int i = b.tid; i < M8; i += NT) vDZ[i] += vCZ[i];
            for (int i = b.tid; i < q; i += NT) vDY[i] += vCY[i];
            Mat::sync(b);
        }
    }
    if (!kBackward) {
        for (int i = b.tid; i < n; i += NT) put_(a.dx, io32, (size_t)qp * n + i, vDX[i]);
        for (int i = b.tid; i < m; i += NT) {
            put_(a.dz, io32, (size_t)qp * m + i, vDZ[i]);
            put_(a.ds, io32, (size_t)qp * m + i, (-(rsg ? rsg[i] : T(0)) - vDZ[i]) * vD[i]);
        }
        for (int i = b.tid; i < q; i += NT) put_(a.dy, io32, (size_t)qp * q + i, vDY[i]);
        return;
    }
    // ---- gradients (qp.py:157-173); a NULL output = that gradient is not wanted (ctx.needs_input_grad)
    const In<T> zhg(a.zhat, (size_t)qp * n, io32), nug(q > 0 ? a.nu : nullptr, (size_t)qp * q, io32);
    for (int i = b.tid; i < n; i += NT) {
        vZH[i] = zhg[i];
        if (a.dp) put_(a.dp, io32, (size_t)qp * n + i, vDX[i]);
    }
    for (int i = b.tid; i < m; i += NT) {
        vLM[i] = lamg[i];
        if (a.dh) put_(a.dh, io32, (size_t)qp * m + i, -vDZ[i]);
    }
    for (int i = b.tid; i < q; i += NT) {
        vNU[i] = nug[i];
        if (a.db) put_(a.db, io32, (size_t)qp * q + i, -vDY[i]);
    }
    // the solution of the backward KKT system itself, for callers that reduce shared-parameter gradients
    // over the batch as one contraction (qpx_batch_outer) instead of B outer products
    if (a.dx) for (int i = b.tid; i < n; i += NT) put_(a.dx, io32, (size_t)qp * n + i, vDX[i]);
    if (a.dz) for (int i = b.tid; i < m; i += NT) put_(a.dz, io32, (size_t)qp * m + i, vDZ[i]);
    if (a.dy) for (int i = b.tid; i < q; i += NT) put_(a.dy, io32, (size_t)qp * q + i, vDY[i]);
    Mat::sync(b);
    if (a.dQ) {
        const size_t o = (size_t)qp * n * n;
        for (int idx = b.tid; idx < n * n; idx += NT) {
            const int r = idx / n, c = idx - r * n;
            put_(a.dQ, io32, o + idx, T(0.5) * (vDX[r] * vZH[c] + vZH[r] * vDX[c]));
        }
    }
    if (a.dG) {
        const size_t o = (size_t)qp * m * n;
        for (int idx = b.tid; idx < m * n; idx += NT) {
            const int r = idx / n, c = idx - r * n;
            put_(a.dG, io32, o + idx, vDZ[r] * vZH[c] + vLM[r] * vDX[c]);
        }
    }
    if (q > 0 && a.dA) {
        const size_t o = (size_t)qp * q * n;
        for (int idx = b.tid; idx < q * n; idx += NT) {
            const int r = idx / n, c = idx - r * n;
            put_(a.dA, io32, o + idx, vDY[r] * vZH[c] + vNU[r] * vDX[c]);
        }
    }
    QPX_PROF(6)
    QPX_PROF_DUMP(F + lay.prof, T)          // (profiling build only: the blob's timer words, scripts/prof_backward.py)
}

template <class T, class Mat, bool kBackward>
QPX_DEV void kkt_mat_body(const Block& b, const KktArgs<T>& a, int qp, T* lds)
{
    typename Mat::Pos g(b);
    g.assign(b, reinterpret_cast<int*>(lds));
    Mat::with_role(g, [&](const auto& gp) { kkt_mat_role<T, Mat, kBackward>(b, a, qp, lds, gp); });
}

template <class T, int GS, int NBL, bool kBackward>
QPX_DEV void kkt_grid_body(const Block& b, const KktArgs<T>& a, int qp, T* lds)
{
    kkt_mat_body<T, GridMat<T, GS, NBL>, kBackward>(b, a, qp, lds);
}

// ------------------------------------------------------------------------------------------
// The finishing stage as ONE kernel (round 4; rounds 2-3 ran it as ~30 host-driven tensor ops per step):
// `steps` iterations of the reference's PDIPM loop (batch.py:92-198: affine + centring-corrector) in the ORIGINAL
// variables (x, s, z, y), started from the iterate the caller passes in, on the format-3 blob.  The KKT residuals
//   rx = Q x + p + G^T z + A^T y,   rz = G x + s - h,   ry = A x - b          (batch.py:93-101)
// are formed from the caller's Q, G, A with double accumulation whatever T is -- the loop kernel iterates on
// pre-computed products (R = G Q^-1 G^T ...) whose float32 rounding error is a perturbation of the PROBLEM that no
// number of loop iterations removes; residuals against the original data do.  Per step: one factorisation of
// T = R + diag(s/z) (ldl_inv, the factor stays in registers), the affine and the corrector solve through the
// condensed inverse (the `apply` of kkt_mat_role), each refined `refine` times on the residual of the original
// KKT system (kkt_resid_reg, batch.py:228-241), step lengths, centring -- and the reference's best-iterate rule
// (batch.py:118-139: strict <, NaN never wins) with the reference's residual ||rx|| + ||rz|| + ||ry|| + nineq mu.
// out[c] = base[c] + sum_r M1[r][c] v1[r] + sum_r M2[r][c] v2[r] + sum_r M3[r][c] v3[r]   (c < cols; M2 / M3 may be null),
// ONE double accumulator per output and one rounding to T at the end: the terms of a KKT residual are large and
// cancel, so a float32 partial sum would bury the residual in its own rounding (thread per column, coalesced rows)
template <class T, class V>
QPX_DEV void resid_cols(const Block& blk, T* out, const T* base, int cols, const T* M1, const V* v1, int r1, const T* M2,
                        const V* v2, int r2, const T* M3, const V* v3, int r3)
{
    for (int c = blk.tid; c < cols; c += blk.nt) {
        double a0 = base ? (double)base[c] : 0.0, a1 = 0, a2 = 0, a3 = 0;
        auto add = [&](const T* M, const V* v, int rows) {
            const T* col = M + c;
            int r = 0;
            for (; r + 4 <= rows; r += 4) {
                const T m0 = col[(size_t)r * cols], m1 = col[(size_t)(r + 1) * cols], m2 = col[(size_t)(r + 2) * cols], m3 = col[(size_t)(r + 3) * cols];
                a0 = fma_((double)m0, (double)v[r], a0);
                a1 = fma_((double)m1, (double)v[r + 1], a1);
                a2 = fma_((double)m2, (double)v[r + 2], a2);
                a3 = fma_((double)m3, (double)v[r + 3], a3);
            }
            for (; r < rows; ++r) a0 = fma_((double)col[(size_t)r * cols], (double)v[r], a0);
        };
        add(M1, v1, r1);
        if (M2) add(M2, v2, r2);
        if (M3) add(M3, v3, r3);
        out[c] = (T)((a0 + a1) + (a2 + a3));
    }
}
// The same sums by ALL nt threads of the workgroup (the thread-per-column form above keeps cols of them busy -- 100 of 256
// at C2 -- each walking every row with four loads in flight): the rows of each matrix are dealt over G = nt / cols groups
// of threads, eight loads in flight per thread, and the groups' partial sums meet in `part` (nt doubles of LDS) in a
// fixed order.  Two barriers inside (`sync`); out may be base.
template <class T, class V, class Sync>
QPX_DEV void resid_cols_all(const Block& blk, int nt, T* out, const T* base, int cols, const T* M1, const V* v1, int r1,
                            const T* M2, const V* v2, int r2, const T* M3, const V* v3, int r3, double* part, Sync&& sync)
{
    const int G = nt / cols;
    if (G < 2) {                                     // more columns than half the threads: the thread-per-column form
        resid_cols<T, V>(blk, out, base, cols, M1, v1, r1, M2, v2, r2, M3, v3, r3);
        sync();
        return;
    }
    const int c = blk.tid % cols, grp = blk.tid / cols;
    if (grp < G) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
        auto add = [&](const T* M, const V* v, int rows) {
            const int lo = (int)((long long)rows * grp / G), hi = (int)((long long)rows * (grp + 1) / G);
            const T* col = M + c;
            int r = lo;
            for (; r + 8 <= hi; r += 8) {
                const T m0 = col[(size_t)r * cols], m1 = col[(size_t)(r + 1) * cols], m2 = col[(size_t)(r + 2) * cols], m3 = col[(size_t)(r + 3) * cols];
                const T m4 = col[(size_t)(r + 4) * cols], m5 = col[(size_t)(r + 5) * cols], m6 = col[(size_t)(r + 6) * cols], m7 = col[(size_t)(r + 7) * cols];
                a0 = fma_((double)m0, (double)v[r], a0);
                a1 = fma_((double)m1, (double)v[r + 1], a1);
                a2 = fma_((double)m2, (double)v[r + 2], a2);
                a3 = fma_((double)m3, (double)v[r + 3], a3);
                a4 = fma_((double)m4, (double)v[r + 4], a4);
                a5 = fma_((double)m5, (double)v[r + 5], a5);
                a6 = fma_((double)m6, (double)v[r + 6], a6);
                a7 = fma_((double)m7, (double)v[r + 7], a7);
            }
            for (; r < hi; ++r) a0 = fma_((double)col[(size_t)r * cols], (double)v[r], a0);
        };
        add(M1, v1, r1);
        if (M2) add(M2, v2, r2);
        if (M3) add(M3, v3, r3);
        part[grp * cols + c] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
    sync();
    if (blk.tid < cols) {
        double sum = base ? (double)base[blk.tid] : 0.0;
        for (int g2 = 0; g2 < G; ++g2) sum += part[g2 * cols + blk.tid];
        out[blk.tid] = (T)sum;
    }
    sync();
}
// out[r] = sign * (b1[r] - b2[r] + sum_c M[r][c] v[c])   (r < rows; b1 / b2 may be null): row dots in double, one rounding
template <class T, class V, class B1>
QPX_DEV void resid_rows(const Block& blk, T* out, const B1* b1, const T* b2, const T* M, const V* v, int rows, int cols, double sign)
{
    constexpr int RB = 4;
    const int lane = blk.lane(), w = blk.uniform(blk.wave()), nw = blk.nwaves();
    for (int r0 = w * RB; r0 < rows; r0 += nw * RB) {
        double acc[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) acc[u] = 0.0;
        for (int c = lane; c < cols; c += kWave) {
            const double x = (double)v[c];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = (r0 + u < rows) ? r0 + u : rows - 1;
                acc[u] = fma_((double)M[(size_t)r * cols + c], x, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) acc[u] = wave_sum(blk, acc[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < RB; ++u)
                if (r0 + u < rows) {
                    const int r = r0 + u;
                    out[r] = (T)(sign * ((b1 ? (double)b1[r] : 0.0) - (b2 ? (double)b2[r] : 0.0) + acc[u]));
                }
        }
    }
}

template <class T, class Mat, class P>
QPX_DEV void polish_mat_role(const Block& b, const PolishArgs<T>& a, int qp, T* lds, const P& g)
{
    constexpr int M8 = Mat::MP, NT = Mat::NT;
    const int n = a.n, m = a.m, q = a.q;
    const FacLayout lay = fac_layout(n, m, q, a.images);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const size_t v = align4(max2(max2((size_t)n, (size_t)M8), (size_t)q));
    T* rd = lds;
    T* vD = rd + v;       // s/z (1 on the pad): the diagonal added to R
    T* vZt = vD + v;      // z in T: the affine step's rs
    T* vRX = vZt + v;     // residuals of the iterate
    T* vRZ = vRX + v;
    T* vRY = vRZ + v;
    T* vRH = vRY + v;     // work vectors of a solve
    T* vTm = vRH + v;
    T* vWX = vTm + v;
    T* vWY = vWX + v;
    T* vCX = vWY + v;
    T* vCZ = vCX + v;
    T* vCY = vCZ + v;
    T* vDZA = vCY + v;    // affine direction
    T* vDSA = vDZA + v;
    T* vDXA = vDSA + v;
    T* vDYA = vDXA + v;
    T* vRSC = vDYA + v;   // corrector right-hand side; then ds of the full step
    T* vDZ = vRSC + v;    // corrector direction, then the full step
    T* vDX = vDZ + v;
    T* vDY = vDX + v;
    T* vZero = vDY + v;
    T* scT = vZero + v;   // 16 elements of T holding 8 doubles' worth of nothing: keeps the doubles behind it 8-byte aligned
    // The iterate and the best iterate live in DOUBLE whatever T is: in float32 the rounding of x alone puts a floor of
    // ~eps32 ||Q|| ||x|| under the residual that ranks the iterates, and the second step could never "win" (measured:
    // the kernel stayed at 1.1e-4 from the float64 answer where the host version, which iterated in float64, reached 2.6e-5)
    double* xd = reinterpret_cast<double*>(scT + 16);
    double* sd = xd + v;
    double* zd = sd + v;
    double* yd = zd + v;
    double* bxd = yd + v;
    double* bsd = bxd + v;
    double* bzd = bsd + v;
    double* byd = bzd + v;
    double* sc = byd + v; // 16 scalars
    double* part = sc + 16;                     // NT partial sums of the column products (resid_cols_all)
    T* scr = reinterpret_cast<T*>(part + NT);   // Mat::scratch_elems()
    enum { kMu = 0, kTot, kBest, kBetter, kAlpha };

    const T* Qg = a.Q + (size_t)qp * a.sQ;
    const T* Gg = a.G + (size_t)qp * a.sG;
    const T* Ag = (q > 0) ? a.A + (size_t)qp * a.sA : nullptr;
    const T* pg = a.p + (size_t)qp * a.sp;
    const T* hg = a.h + (size_t)qp * a.sh;
    const T* bg = (q > 0) ? a.b + (size_t)qp * a.sb : nullptr;
    const int lane = b.lane();
    const bool w0 = g.lead(b);
    const double mD = (double)m;
    const double tiny = (double)Lim<T>::tiny();

    for (int i = b.tid; i < (int)v; i += NT) {
        xd[i] = (i < n) ? (double)a.zhat[(size_t)qp * n + i] : 0.0;
        sd[i] = (i < m) ? (double)a.slack[(size_t)qp * m + i] : 1.0;
        zd[i] = (i < m) ? (double)a.lam[(size_t)qp * m + i] : 1.0;
        yd[i] = (i < q) ? (double)a.nu[(size_t)qp * q + i] : 0.0;
        bxd[i] = xd[i]; bsd[i] = sd[i]; bzd[i] = zd[i]; byd[i] = yd[i];
        vZero[i] = T(0);
        vDZA[i] = vDSA[i] = vDXA[i] = vDYA[i] = vRSC[i] = vDZ[i] = vDX[i] = vDY[i] = T(0);
        vRX[i] = vRZ[i] = vRY[i] = vRH[i] = vCX[i] = vCZ[i] = vCY[i] = vWX[i] = vWY[i] = vZt[i] = T(0);
    }
    if (b.tid == 0) sc[kBest] = __builtin_huge_val();
    Mat::sync(b);

    typename Mat::Regs E;
    bool ok = true;
    // one application of the condensed KKT inverse with the factor in E (kkt_mat_role::apply)
    auto wgsync = [&]() { Mat::sync(b); };
    const T* nullT = nullptr;
    auto apply = [&](const T* rX, const T* rY, T* rH, T* oZ, T* oX, T* oY) {
        resid_cols_all<T, T>(b, NT, rH, rH, m, F + lay.MT, rX, n, nullT, nullT, 0, nullT, nullT, 0, part, wgsync);      // rH += M rX
        if (q > 0) {
            for (int j = b.tid; j < m; j += NT) {
                T acc = rH[j];
                for (int r = 0; r < q; ++r) acc = fma_(F[lay.W + (size_t)j * q + r], rY[r], acc);
                rH[j] = acc;
            }
            Mat::sync(b);
        }
        Mat::solve_neg(b, g, E, rd, m, rH, oZ, vTm, scr);
        resid_cols_all<T, T>(b, NT, oX, nullT, n, F + lay.Kneg, rX, n, nullT, nullT, 0, nullT, nullT, 0, part, wgsync);  // oX = -K rX
        block_matvec16<T, 2>(b, oX, F + lay.MT, oZ, n, m);
        if (q > 0) {
            Mat::sync(b);
            block_matTvec<T, 1>(b, oX, F + lay.NTn, rY, q, n);
            for (int r = b.tid; r < q; r += NT) {
                T acc = 0;
                for (int c2 = 0; c2 < q; ++c2) acc = fma_(F[lay.S11i + (size_t)r * q + c2], rY[c2], acc);
                for (int k = 0; k < n; ++k) acc = fma_(F[lay.NTn + (size_t)r * n + k], rX[k], acc);
                for (int j = 0; j < m; ++j) acc = fma_(-F[lay.W + (size_t)j * q + r], oZ[j], acc);
                oY[r] = acc;
            }
        }
        Mat::sync(b);
    };
    // solve_kkt(rX, rS, rZ, rY) -> (oX, oZ, oY) [ds = (-rS - oZ) s/z is the caller's], refined a.refine times on the
    // residual of the original system:  resx = Q dx + G^T dz + A^T dy + rx,  resz = G dx + ds + rz,  resy = A dx + ry
    auto solve = [&](const T* rX, const T* rS, const T* rZ, const T* rY, T* oZ, T* oX, T* oY) {
        for (int i = b.tid; i < M8; i += NT) vRH[i] = (i < m) ? rS[i] * vD[i] - rZ[i] : T(0);
        Mat::sync(b);
        apply(rX, rY, vRH, oZ, oX, oY);
        for (int it = 0; it < a.refine; ++it) {
            // resx = rx + Q dx + G^T dz + A^T dy;  -resz = -(ds + rz + G dx);  resy = ry + A dx
            for (int i = b.tid; i < M8; i += NT) vCZ[i] = (i < m) ? (-rS[i] - oZ[i]) * vD[i] + rZ[i] : T(0);       // ds + rz
            Mat::sync(b);
            resid_cols_all<T, T>(b, NT, vWX, rX, n, Qg, oX, n, Gg, oZ, m, Ag, oY, q, part, wgsync);
            resid_rows<T, T, T>(b, vRH, vCZ, nullptr, Gg, oX, m, n, -1.0);
            if (Ag) resid_rows<T, T, T>(b, vWY, rY, nullptr, Ag, oX, q, n, 1.0);
            for (int i = b.tid + m; i < M8; i += NT) vRH[i] = T(0);
            Mat::sync(b);
            apply(vWX, vWY, vRH, vCZ, vCX, vCY);
            for (int i = b.tid; i < n; i += NT) oX[i] += vCX[i];
            for (int i = b.tid; i < M8; i += NT) oZ[i] += vCZ[i];
            for (int i = b.tid; i < q; i += NT) oY[i] += vCY[i];
            Mat::sync(b);
        }
    };
    // min over dv < 0 of -v / dv (inf if none), by the lead wave
    auto step_len = [&](const double* vv, const T* dv) {
        double al = __builtin_huge_val();
        for (int i = lane; i < m; i += kWave) {
            const double d = (double)dv[i];
            if (d < 0.0) al = min2_(al, -vv[i] / d);
        }
        return wave_min(b, al);
    };

    for (int st = 0; st <= a.steps; ++st) {
        // ---- residuals of the current iterate (double accumulation), mu, the reference's total residual, best iterate
        resid_cols_all<T, double>(b, NT, vRX, pg, n, Qg, xd, n, Gg, zd, m, Ag, yd, q, part, wgsync);         // Q symmetric: column-parallel over its rows
        resid_rows<T, double, double>(b, vRZ, sd, hg, Gg, xd, m, n, 1.0);
        if (Ag) resid_rows<T, double, double>(b, vRY, (const double*)nullptr, bg, Ag, xd, q, n, 1.0);
        Mat::sync(b);
        if (w0) {
            double sz = 0, nx = 0, nz = 0, ny = 0;
            for (int i = lane; i < m; i += kWave) { sz = fma_(sd[i], zd[i], sz); nz = fma_((double)vRZ[i], (double)vRZ[i], nz); }
            for (int i = lane; i < n; i += kWave) nx = fma_((double)vRX[i], (double)vRX[i], nx);
            for (int i = lane; i < q; i += kWave) ny = fma_((double)vRY[i], (double)vRY[i], ny);
            sz = wave_sum(b, sz); nx = wave_sum(b, nx); nz = wave_sum(b, nz); ny = wave_sum(b, ny);
            const double mu = abs_(sz) / mD;
            const double tot = sqrt_(nx) + sqrt_(nz) + sqrt_(ny) + mD * mu;
            const bool better = tot < sc[kBest];                     // false for NaN: a non-finite iterate never wins
            b.wave_sync();
            if (lane == 0) {
                sc[kMu] = mu; sc[kTot] = tot;
                sc[kBetter] = better ? 1.0 : 0.0;
                if (better) sc[kBest] = tot;
            }
        }
        Mat::sync(b);
        if (sc[kBetter] != 0.0 && st > 0) {
            for (int i = b.tid; i < (int)v; i += NT) { bxd[i] = xd[i]; bsd[i] = sd[i]; bzd[i] = zd[i]; byd[i] = yd[i]; }
        }
        if (st == a.steps) break;
        // ---- factor T = R + diag(s/z)   (d = z/s clamped away from 0 as in the host version: batch.py:146)
        for (int i = b.tid; i < M8; i += NT) {
            double d = 1.0;
            if (i < m) d = max2_(sd[i], tiny) / max2_(zd[i], tiny);
            vD[i] = (T)d;
            vZt[i] = (i < m) ? (T)zd[i] : T(0);
        }
        Mat::sync(b);
        Mat::load(b, g, E, Mat::image(F, lay));
        Mat::add_diag(g, E, vD);
        ok = Mat::ldl_inv(b, g, E, scr, rd, m);
        if (!ok) break;                                              // uniform
        // ---- affine direction: solve_kkt(rx, rs = z, rz, ry)   (batch.py:160-162)
        solve(vRX, vZt, vRZ, vRY, vDZA, vDXA, vDYA);
        for (int i = b.tid; i < M8; i += NT) vDSA[i] = (i < m) ? (-vZt[i] - vDZA[i]) * vD[i] : T(0);
        Mat::sync(b);
        if (w0) {
            double al = min2_(step_len(zd, vDZA), step_len(sd, vDSA));
            al = min2_(al, 1.0);
            double t3 = 0, sz = 0;
            for (int i = lane; i < m; i += kWave) {
                t3 = fma_(sd[i] + al * (double)vDSA[i], zd[i] + al * (double)vDZA[i], t3);
                sz = fma_(sd[i], zd[i], sz);
            }
            t3 = wave_sum(b, t3); sz = wave_sum(b, sz);
            double sig = t3 / sz;
            sig = sig * sig * sig;                                   // batch.py:168
            const double mu = sc[kMu];
            for (int i = lane; i < m; i += kWave)
                vRSC[i] = (T)((-mu * sig + (double)vDSA[i] * (double)vDZA[i]) / max2_(sd[i], tiny));       // batch.py:171
        }
        Mat::sync(b);
        // ---- centring-corrector direction: solve_kkt(0, rs_cor, 0, 0)
        solve(vZero, vRSC, vZero, vZero, vDZ, vDX, vDY);
        for (int i = b.tid; i < (int)v; i += NT) {
            const T dsc = (i < m) ? (-vRSC[i] - vDZ[i]) * vD[i] : T(0);
            vDZ[i] = (i < m) ? vDZA[i] + vDZ[i] : T(0);
            vRSC[i] = (i < m) ? vDSA[i] + dsc : T(0);                // ds of the full step
            vDX[i] = (i < n) ? vDXA[i] + vDX[i] : T(0);
            vDY[i] = (i < q) ? vDYA[i] + vDY[i] : T(0);
        }
        Mat::sync(b);
        if (w0) {
            double al = 0.999 * min2_(step_len(zd, vDZ), step_len(sd, vRSC));                     // batch.py:193
            al = min2_(al, 1.0);
            b.wave_sync();
            if (lane == 0) sc[kAlpha] = al;
        }
        Mat::sync(b);
        {
            const double al = sc[kAlpha];
            for (int i = b.tid; i < (int)v; i += NT) {
                if (i < n) xd[i] = fma_(al, (double)vDX[i], xd[i]);
                if (i < m) { sd[i] = fma_(al, (double)vRSC[i], sd[i]); zd[i] = fma_(al, (double)vDZ[i], zd[i]); }
                if (i < q) yd[i] = fma_(al, (double)vDY[i], yd[i]);
            }
        }
        Mat::sync(b);
    }
    if (!ok && b.tid == 0 && a.status) a.status[qp] |= QPX_ST_KKT_BREAKDOWN;
    for (int i = b.tid; i < n; i += NT) a.zhat[(size_t)qp * n + i] = (T)bxd[i];
    for (int i = b.tid; i < m; i += NT) { a.lam[(size_t)qp * m + i] = (T)bzd[i]; a.slack[(size_t)qp * m + i] = (T)bsd[i]; }
    for (int i = b.tid; i < q; i += NT) a.nu[(size_t)qp * q + i] = (T)byd[i];
    if (b.tid == 0 && a.best_resid) a.best_resid[qp] = (T)sc[kBest];
}

template <class T, class Mat>
QPX_DEV void polish_mat_body(const Block& b, const PolishArgs<T>& a, int qp, T* lds)
{
    typename Mat::Pos g(b);
    g.assign(b, reinterpret_cast<int*>(lds));
    Mat::with_role(g, [&](const auto& gp) { polish_mat_role<T, Mat>(b, a, qp, lds, gp); });
}

template <class T, int GS, int NBL>
QPX_DEV void polish_grid_body(const Block& b, const PolishArgs<T>& a, int qp, T* lds)
{
    polish_mat_body<T, GridMat<T, GS, NBL>>(b, a, qp, lds);
}

// LDS elements (of T, tsize bytes each) of the finishing kernel: 22 vectors + 16 of T, 8 vectors + 16 scalars + 256 partial sums of double,
// the matrix operations' scratch
QPX_LAYOUT_HD size_t lds_elems_polish_mat(size_t mp, size_t scratch, int n, int q, size_t tsize)
{
    const size_t v = align4(max2(max2((size_t)n, mp), (size_t)q));
    return 22 * v + 16 + (8 * v + 16 + 256) * (8 / tsize) + scratch;        // (+ 256 doubles: the partial sums of the column products)
}
QPX_LAYOUT_HD size_t lds_elems_polish_grid(int gs, int nbl, int n, int q, size_t tsize)
{
    const size_t mg = (size_t)gs * nbl;
    return lds_elems_polish_mat(mg, 2 * mg + 4 + (size_t)nbl * gs * gs, n, q, tsize);
}

// LDS elements of the KKT-solve / backward kernel: 12 vectors + the matrix operations' scratch
QPX_LAYOUT_HD size_t lds_elems_kkt_mat(size_t mp, size_t scratch, int n, int q)
{
    const size_t v = align4(max2(max2((size_t)n, mp), (size_t)q));
    return 15 * v + scratch;
}
QPX_LAYOUT_HD size_t lds_elems_kkt_grid(int gs, int nbl, int n, int q)
{
    const size_t mg = (size_t)gs * nbl;
    return lds_elems_kkt_mat(mg, 2 * mg + 4 + (size_t)nbl * gs * gs, n, q);
}

}  // namespace qpx
