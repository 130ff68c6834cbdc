// qpx_big.h -- the LARGE-QP kernel family (BASELINE.json configs[3]: batch 128, nz = nineq = 500).
//
// At this size one QP's matrices are 2 MB each: they live in HBM / L2, a QP is worked on by MANY workgroups,
// and the dense contractions are real GEMMs -- so the forward is a stream-ordered SEQUENCE of batched kernels
// (every launch covers the whole batch; no host synchronisation), not one kernel per QP:
//
//   pre-factorisation (pre_factor_kkt, batch.py:375-429), neq = 0:
//     Lq   = chol(Q)                      blocked right-looking Cholesky: panel kernel + MFMA trailing update
//     Zt   = G Lq^-T   (nineq x nz)       blocked triangular solve: panel kernel + MFMA trailing update
//     R    = Zt Zt^T = G Q^-1 G^T         one MFMA GEMM (the reference's R, batch.py:396-399)
//   PDIPM loop (batch.py:47-207), the condensed iteration of ipm_loop_body (qpx_grid.h) with its vectors in HBM:
//     per pass: R z' (row-dot kernel), wave-0 vector work (phase kernels), T = R + diag(s/z) factored by the
//     same blocked Cholesky, two solves = four blocked triangular substitutions
//   backward (qp.py:127-182): one factorisation + one solve + the outer products.
//
// Blocks are 64 x 64; every matrix is stored square, row-major, padded to a multiple of 64 (identity padding on
// the diagonal).  A factor keeps L in the lower triangle and L^T in the upper one (both substitutions then read
// ROWS, coalesced) and the inverses of its diagonal blocks W_kk = L_kk^-1 beside it: a diagonal-block solve is a
// 64 x 64 mat-vec, and the panel operation X W_kk^T has independent outputs (four threads per row).
// The trailing updates C -= A B^T run on v_mfma_f64_16x16x4 (f32: v_mfma_f32_16x16x4): a workgroup owns a 64 x 64
// tile of C, each of its four waves a 32 x 32 quadrant (four accumulator tiles), operands staged through LDS.
#pragma once
#include <type_traits>

#include "qpx_grid.h"
#include "qpx_tile.h"

namespace qpx {

constexpr int kBB = 64;          // block order
// (round 4) Every matrix of the blob is stored BLOCK by BLOCK: block (rb, cb) of a matrix with `ld` elements per row is
// the 4096 contiguous elements at ((rb * (ld / 64)) + cb) * 4096, row-major inside.  A workgroup that reads a 64 x 64
// block -- every GEMM operand, every substitution step -- then reads 32 KB of consecutive memory instead of 64 pieces of
// 512 bytes a matrix row apart: the family is HBM-bound (profiles/archive/r04c_c4_pmc_*: the whole pass moves 1.9 GB at an
// effective 3.5 TB/s), and the DRAM pages like the long runs better.
constexpr int kBE = kBB * kBB;   // elements of a block
QPX_LAYOUT_HD size_t big_blk(int ld, int rb, int cb) { return ((size_t)rb * (ld / kBB) + cb) * kBE; }
QPX_LAYOUT_HD size_t big_at(int ld, int i, int j) { return big_blk(ld, i >> 6, j >> 6) + (size_t)(i & 63) * kBB + (j & 63); }
constexpr int kDiagNoWt = 4;         // `tile` bit of the diagonal-block eliminations: do not store W_kk^T
constexpr int kBigPolVecs = 32;      // element-sized vectors of the finishing stage's region (BigLayout::pol)
constexpr int kMaxSide = 3;          // side streams the host may spread the parts of a batch over (qpx_api.inc: big_split)
constexpr int kPoolSlots = 2 * kMaxSide + 1;      // ... + one helper stream per part (R z' beside the factorisation)
constexpr int kBL = kBB + 2;     // LDS row stride of a staged block (elements): 2-way bank conflicts at most
QPX_LAYOUT_HD int big_pad(int x) { return (x + kBB - 1) / kBB * kBB; }

// per-QP vectors of the loop (each VP = max(NP, MP, QP) elements), same roles as in ipm_loop_body; the last five serve
// the equality constraints: bvBQ = b (zero padded), bvTB = L11^-1 b (KKT solve: L11^-1 ry), bvT1 = S11^-1 (b + Yt u)
// (KKT solve: S11^-1 (ry - Yt u_x)), bvNU = work vector of nu / dy, bvPQ = 0 on [0, q), 1 on the pad of S11's diagonal
enum BigVec {
    bvP, bvU, bvC, bvR1, bvZ, bvS, bvA, bvB, bvD, bvBZ, bvBS, bvRH, bvX, bvDZA, bvDSA, bvRSC, bvRZ, bvRS, bvW, bvY,
    bvONE, bvBQ, bvTB, bvT1, bvNU, bvPQ, bvCount
};
enum BigScal { bsTau = 0, bsBtau, bsSigz, bsSigs, bsBres, bsFeasPrev, bsAlphaPrev, bsMu, bsSzdot, bsGt1 = 15 };
enum BigCtrl { bcStop = 0, bcNnot, bcFloor, bcSt, bcIters, bcFail };

// Equality constraints (round 4).  With Lq = chol(Q), Yt = A Lq^-T (neq x nz), S11 = Yt Yt^T = A Q^-1 A^T = L11 L11^T:
//   K = Q^-1 - Q^-1 A^T S11^-1 A Q^-1 = Lq^-T P Lq^-1,   P = I - Yt^T S11^-1 Yt   (the projector on null(Yt)),
// so everything the loop needs keeps its form with Zt replaced by the PROJECTED Ztp = Zt P = Zt - (Vh L11^-1) Yt,
// Vh = Zt Yt^T L11^-T (nineq x neq):  R = Ztp Ztp^T (batch.py:396-399,424),  M = G K = Ztp Lq^-1,  W = G N = Vh L11^-1.
// The blob keeps Yt (right behind Zt, same row length: the solve with Lq^-T runs once over the stacked rows of G and A),
// L11 with its diagonal-block inverses, and Vh; Us is the scratch of the pre-factorisation (Vh L11^-1).
struct BigLayout {
    size_t Lq, Wq, Zt, Yt, R, T, Wt, S11, Wy, Vh, Us, vec, scal, ctrl, pol, total;
    int NP, MP, QP, VP;
    QPX_LAYOUT_HD size_t v(int i) const { return vec + (size_t)i * VP; }
};
QPX_LAYOUT_HD BigLayout big_layout(int n, int m, int q = 0)
{
    BigLayout L;
    L.NP = big_pad(n); L.MP = big_pad(m); L.QP = q > 0 ? big_pad(q) : 0; L.VP = L.NP > L.MP ? L.NP : L.MP;
    size_t o = 0;
    L.Lq = o;  o += (size_t)L.NP * L.NP;
    L.Wq = o;  o += (size_t)(L.NP / kBB) * 2 * kBB * kBB;      // per diagonal block: W_kk, then W_kk^T
    L.Zt = o;  o += (size_t)L.MP * L.NP;
    L.Yt = o;  o += (size_t)L.QP * L.NP;
    L.R = o;   o += (size_t)L.MP * L.MP;
    L.T = o;   o += (size_t)L.MP * L.MP;
    L.Wt = o;  o += (size_t)(L.MP / kBB) * 2 * kBB * kBB;
    L.S11 = o; o += (size_t)L.QP * L.QP;
    L.Wy = o;  o += (size_t)(L.QP / kBB) * 2 * kBB * kBB;
    L.Vh = o;  o += (size_t)L.MP * L.QP;
    L.Us = o;  o += (size_t)L.MP * L.QP;
    L.vec = o; o += (size_t)bvCount * L.VP;
    L.scal = o; o += 16;
    L.ctrl = o; o += 16;
    // the finishing stage (qpx_big_polish.h): iterate and best iterate in DOUBLE whatever the element type -- room for 16
    // vectors of VP doubles when the elements are floats (32 of them when they are doubles)
    L.pol = o; o += (size_t)kBigPolVecs * L.VP;
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------ pack
template <class T> struct BigPackArgs {
    int B, rows, cols, P, ldp, sym;       // source rows x cols -> destination P x ldp (sym: identity on the padded diagonal)
    const T* src; long long ssrc;
    T* dst; size_t sdst;
    int io32;                             // T = double: the source is a float32 array (QPX_F32_WIDE)
};
// grid (B, P / 16): 16 destination rows per workgroup
template <class T> QPX_DEV void big_pack_body(const Block& b, const BigPackArgs<T>& a, int qp, int chunk)
{
    const In<T> S(a.src, (size_t)qp * a.ssrc, a.io32);
    T* D = a.dst + (size_t)qp * a.sdst;
    for (int e = b.tid; e < 16 * a.ldp; e += b.nt) {
        const int i = 16 * chunk + e / a.ldp, j = e % a.ldp;
        if (i >= a.P) continue;
        T v = T(0);
        // (sym: Q is copied as it is -- the reference assumes a symmetric Q too, qp.py:81-85 -- a symmetrising read of
        // Q[j][i] is a strided one: 0.34 ms instead of 0.05 for the 128 matrices of C4)
        if (i < a.rows && j < a.cols) v = S[(size_t)i * a.cols + j];
        else if (a.sym && i == j) v = T(1);
        D[big_at(a.ldp, i, j)] = v;
    }
}

// ------------------------------------------------------------------------------------------ diagonal block
// The diagonal block D = M[k][k] of a panel step (Ms + diag at the first step of T = R + diag), one workgroup per QP:
// the 64 x 64 block is held by a 16 x 16 thread grid in registers and eliminated by grid_ldl_inv (qpx_grid.h, the
// routine of the thread-grid loop kernels): D = L~ diag(d) L~^T with W~ = L~^-1 built in place, 1/d in LDS.  Written:
//   W_kk = L_kk^-1 = diag(d^-1/2) W~   (L_kk = L~ diag(d^1/2) is the Cholesky factor)   and its transpose,
// 2 x 64 x 64 elements per block.  Nothing downstream reads the diagonal block of the factor itself: the panel
// below it becomes X W_kk^T (a GEMM, big_gemm_body) and the substitutions multiply by W_kk / W_kk^T.
template <class T> struct BigPanelArgs {
    int B, k;
    const T* M; size_t sM; int ld;        // source of the block (T, or R at the first step)
    const T* dg; size_t sdg;              // first-step diagonal (added) or null
    T* W; size_t sW;                      // W blocks out
    int* ctrl; size_t sctrl;              // per-QP control words (elements of int); may be null
    int fail_bit;                         // status bit OR-ed into ctrl[bcFail] when a pivot breaks down
    int check_stop;                       // skip QPs whose ctrl[bcStop] is set (loop factorisations)
    int tile;                             // f64: eliminate on the matrix cores (big_diag_block_tile)
};
// LDS of a diagonal-block elimination: thread-grid form 3 * 64 + 8; matrix-core form (f64) the staged result
// (64 x 66) + the tile routine's scratch + 64 reciprocal pivots
QPX_LAYOUT_HD size_t big_diag_scratch_elems() { return TileMat<kBB / 16, 4, true>::scratch_elems() + kBB + 8; }     // the larger of the two matrix-core forms
QPX_LAYOUT_HD size_t big_panel_lds_elems() { return (size_t)kBB * kBL + big_diag_scratch_elems(); }

// D(i, j) = element (i, j) of the 64 x 64 block (any source: global memory, or the LDS tile a trailing update has just
// finished); W: the 2 x 64 x 64 output of the block; lds: 3 * 64 + 8 elements of scratch.  All 256 threads.
template <class T, class Elem>
QPX_DEV void big_diag_block_grid(const Block& b, Elem&& D, T* W, int* ctrl, int fail_bit, T* lds, bool wt = true)
{
    constexpr int GS = 16, NBL = kBB / GS;
    const GridPos<GS> g(b);
    T* vec2 = lds;                // 2 x 64: published column / row of a pivot step
    T* dsl = vec2 + 2 * kBB;      // pivots
    T* rd = dsl + 8;              // 1 / d_k
    T E[gtri(NBL)];
#pragma unroll
    for (int li = 0; li < NBL; ++li)
#pragma unroll
        for (int lj = 0; lj <= li; ++lj) E[gidx(li, lj)] = D(GS * li + g.a, GS * lj + g.b);
    GridPos<GS>::sync(b);
    const bool ok = grid_ldl_inv<T, GS, NBL>(b, g, E, vec2, dsl, rd, kBB);
    if (!ok) {
        if (b.tid == 0 && ctrl) ctrl[bcFail] |= fail_bit;
        return;
    }
    T* Wt = W + kBB * kBB;
#pragma unroll
    for (int li = 0; li < NBL; ++li)
#pragma unroll
        for (int lj = 0; lj < NBL; ++lj) {
            const int i = GS * li + g.a, j = GS * lj + g.b;
            T v = T(0);
            if (lj <= li) {
                const T rs = sqrt_(rd[i]);
                v = (j < i) ? E[gidx(li, lj)] * rs : (j == i ? rs : T(0));
            }
            W[i * kBB + j] = v;
            if (wt) Wt[j * kBB + i] = v;
        }
}

// The same elimination on the matrix cores (f64): ONE wave holds the lower triangle as ten 16 x 16 accumulator tiles
// and runs the rank-4 blocked ldl_inv of the tile loop kernel (qpx_tile.h, TileMat<4, 1>: 16 panels, wave-level
// ordering only, no workgroup barrier per pivot); the other waves zero the upper tiles of the staged result and
// then help to write W and W^T out coalesced.  stage: 64 x kBL elements of LDS -- MAY be the array D reads (wave 0
// is done with D before it writes, the other waves only touch tiles above the diagonal, which D is never asked
// for); scr: big_diag_scratch_elems().  All 256 threads.
template <class Elem>
QPX_DEV void big_diag_block_tile(const Block& b, Elem&& D, double* W, int* ctrl, int fail_bit, double* stage, double* scr, bool wt = true)
{
    using TM = TileMat<kBB / 16, 1>;
    double* rd = scr + TM::scratch_elems();
    double* flag = rd + kBB;
    const int w = b.uniform(b.wave());
    if (w == 0) {
        const typename TM::Pos p(b);
        typename TM::Regs E;
#pragma unroll
        for (int pp = 0; pp < TM::NPOS; ++pp) {
            const int I = TM::rowof(pp, 0);
#pragma unroll
            for (int J = 0; J < TM::psize(pp); ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) E.e[TM::slot(pp, J)][r] = D(16 * I + p.g + 4 * r, 16 * J + p.c);
        }
        const bool ok = TM::ldl_inv(b, p, E, scr, rd, kBB);
        if (p.lane == 0) flag[0] = ok ? 1.0 : 0.0;
        if (ok) {
#pragma unroll
            for (int pp = 0; pp < TM::NPOS; ++pp) {
                const int I = TM::rowof(pp, 0);
#pragma unroll
                for (int J = 0; J < TM::psize(pp); ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * I + p.g + 4 * r, j = 16 * J + p.c;
                        const double rs = sqrt_(rd[i]);
                        stage[i * kBL + j] = (j < i) ? E.e[TM::slot(pp, J)][r] * rs : (j == i ? rs : 0.0);
                    }
            }
        }
    } else {
        // strictly upper tiles (I < J): 6 tiles x 256 entries over 192 threads
        for (int e = b.tid - kWave; e < 6 * 256; e += b.nt - kWave) {
            const int t = e >> 8, o = e & 255;
            const int I = t < 3 ? 0 : (t < 5 ? 1 : 2), J = t < 3 ? t + 1 : (t < 5 ? t - 1 : 3);
            stage[(16 * I + (o >> 4)) * kBL + 16 * J + (o & 15)] = 0.0;
        }
    }
    b.sync();
    if (flag[0] == 0.0) {
        if (b.tid == 0 && ctrl) ctrl[bcFail] |= fail_bit;
        return;
    }
    double* Wt = W + kBB * kBB;
    for (int e = b.tid; e < kBB * kBB; e += b.nt) {
        const int i = e >> 6, j = e & 63;
        W[e] = stage[i * kBL + j];
        if (wt) Wt[e] = stage[j * kBL + i];
    }
}

// The matrix-core elimination by all FOUR waves in the chain-wave form of the tile loop kernels (qpx_tile.h, TileMat<4, 4,
// true>, the C3 loop kernel's factorisation): wave 0 eliminates the 16 x 16 pivot blocks a panel ahead of the three
// waves that hold the ten tiles -- ~8 us per block instead of ~14 (round 4).  stage may again be the array D reads:
// every lower tile is read and later written by the one wave that owns it, the chain wave zeroes the tiles above
// the diagonal, which D is never asked for.
template <class Elem>
QPX_DEV void big_diag_block_chain(const Block& b, Elem&& D, double* W, int* ctrl, int fail_bit, double* stage, double* scr, bool wt = true)
{
    using TM = TileMat<kBB / 16, 4, true>;
    double* rd = scr + TM::scratch_elems();
    double* flag = rd + kBB;
    const typename TM::Pos p0(b);
    TM::with_role(p0, [&](const auto& gp) {
        using P = typename std::decay<decltype(gp)>::type;
        const P p = gp.fresh();
        typename TM::Regs E;
#pragma unroll
        for (int pp = 0; pp < TM::NPOS; ++pp) {
            const int I = p.row(pp);
#pragma unroll
            for (int J = 0; J < TM::psize(pp); ++J) {
                if (J <= I) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) E.e[TM::slot(pp, J)][r] = D(16 * I + p.g + 4 * r, 16 * J + p.c);
                }
            }
        }
        b.sync();                                    // (D may live where the factorisation's scratch does not, but its reads end here)
        const bool ok = TM::ldl_inv(b, p, E, scr, rd, kBB);
        b.sync();
        if (p.is_chain()) {
            if (p.lane == 0) flag[0] = ok ? 1.0 : 0.0;
            // strictly upper tiles (I < J): 6 tiles x 256 entries by this wave
            for (int e = p.lane; e < 6 * 256; e += kWave) {
                const int t = e >> 8, o = e & 255;
                const int I = t < 3 ? 0 : (t < 5 ? 1 : 2), J = t < 3 ? t + 1 : (t < 5 ? t - 1 : 3);
                stage[(16 * I + (o >> 4)) * kBL + 16 * J + (o & 15)] = 0.0;
            }
        } else if (ok) {
#pragma unroll
            for (int pp = 0; pp < TM::NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int J = 0; J < TM::psize(pp); ++J) {
                    if (J <= I) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * I + p.g + 4 * r, j = 16 * J + p.c;
                            const double rs = sqrt_(rd[i]);
                            stage[i * kBL + j] = (j < i) ? E.e[TM::slot(pp, J)][r] * rs : (j == i ? rs : 0.0);
                        }
                    }
                }
            }
        }
        b.sync();
    });
    if (flag[0] == 0.0) {
        if (b.tid == 0 && ctrl) ctrl[bcFail] |= fail_bit;
        return;
    }
    double* Wt = W + kBB * kBB;
    for (int e = b.tid; e < kBB * kBB; e += b.nt) {
        const int i = e >> 6, j = e & 63;
        W[e] = stage[i * kBL + j];
        if (wt) Wt[e] = stage[j * kBL + i];
    }
}

// tile: 1 / 2 = the matrix-core form by one wave / by four waves in the chain-wave form, when T is double (stage / scr
// as above); otherwise the thread-grid form with scr as its scratch
template <class T, class Elem>
QPX_DEV void big_diag_block(const Block& b, Elem&& D, T* W, int* ctrl, int fail_bit, T* stage, T* scr, int tile)
{
    // tile bit 2 (kDiagNoWt): W_kk alone -- nothing reads the transposed copy of this factor's blocks (round 5: the
    // substitutions take both directions from the rows of W_kk; only the factor of S11 still wants W_kk^T)
    const bool wt = !(tile & kDiagNoWt);
    tile &= 3;
    if constexpr (std::is_same<T, double>::value) {
        if (tile == 2) {
            big_diag_block_chain(b, D, W, ctrl, fail_bit, stage, scr, wt);
            return;
        }
        if (tile) {
            big_diag_block_tile(b, D, W, ctrl, fail_bit, stage, scr, wt);
            return;
        }
    }
    big_diag_block_grid<T>(b, D, W, ctrl, fail_bit, scr, wt);
}

template <class T> QPX_DEV void big_panel_body(const Block& b, const BigPanelArgs<T>& a, int qp, T* lds)
{
    int* ctrl = a.ctrl ? a.ctrl + (size_t)qp * a.sctrl : nullptr;
    if (a.check_stop && ctrl && ctrl[bcStop]) return;
    const T* M = a.M + (size_t)qp * a.sM + big_blk(a.ld, a.k, a.k);      // the diagonal block: 4096 consecutive elements
    const int k0 = a.k * kBB;
    const T* dg = a.dg ? a.dg + (size_t)qp * a.sdg + k0 : nullptr;
    T* W = a.W + (size_t)qp * a.sW + (size_t)a.k * 2 * kBB * kBB;
    if (std::is_same<T, double>::value && (a.tile & 3)) {
        // the matrix-core form reads the block with one wave: bring it into LDS with all four first (coalesced rows)
        for (int e = b.tid; e < kBB * kBB; e += b.nt) {
            const int i = e >> 6, j = e & 63;
            T v = M[e];
            if (dg && i == j) v += dg[i];
            lds[i * kBL + j] = v;
        }
        b.sync();
        big_diag_block<T>(b, [&](int i, int j) { return lds[i * kBL + j]; }, W, ctrl, a.fail_bit, lds, lds + kBB * kBL, a.tile);
        return;
    }
    big_diag_block<T>(b, [&](int i, int j) {
        T v = M[i * kBB + j];
        if (dg && i == j) v += dg[i];
        return v;
    }, W, ctrl, a.fail_bit, lds, lds + kBB * kBL, a.tile & kDiagNoWt);
}

// ------------------------------------------------------------------------------------------ GEMM tile
// C[tile (ti, tj)] = (Cs or C or 0)[tile] (+ diag) + alpha * sum over kb < nk of A[arb0+ti][akb0+kb] B[brb0+tj][bkb0+kb]^T
// grid (B, nti * ntj); lower: tiles with cb0 + ... row block < column block are skipped; mirror: the tile is also
// written transposed (symmetric result stored in full).
template <class T> struct BigGemmArgs {
    int B, nti, ntj, crb0, ccb0, arb0, brb0, akb0, bkb0, nk, lower, mirror, zero_init;
    T* C; size_t sC; int ldc;
    const T* Cs; size_t sCs; int ldcs;
    const T* dg; size_t sdg;
    const T* A; size_t sA; int lda;
    const T* Bm; size_t sB; int ldb;
    T alpha;
    int* ctrl; size_t sctrl; int check_stop;
    // fuse: the workgroup of tile (0, 0) -- the diagonal block the NEXT panel starts from -- goes on to eliminate it
    // (big_diag_block) as soon as its update is done: W block index fuse_k of W, failure bit as in BigPanelArgs
    int fuse, fuse_k, fail_bit, tile;
    T* W; size_t sW;
    int no_swizzle;              // A/B: plain (qp, tile) grid instead of the XCD-aware one (launcher only)
    int transb;                  // Bm is given as [k][column] (the product is A Bm, not A Bm^T): rows bkb0.. of Bm, columns of block brb0 + tj (pipelined form only)
};
// pipelined form (big_gemm2_body): k-chunks of 16, operands row-major [row][k] with a row stride of kGL elements,
// two buffers each; a fused launch also needs the staged tile + the diagonal-block scratch (big_diag_block)
constexpr int kGC = 16;
template <class T> QPX_LAYOUT_HD constexpr int big_gl() { return sizeof(T) == 8 ? 18 : 20; }
template <class T> QPX_LAYOUT_HD size_t big_gemm2_lds_elems(bool fused, bool mirror)
{
    const size_t pipe = (size_t)4 * kBB * big_gl<T>();
    const size_t stage = (size_t)kBB * kBL;
    size_t e = pipe;
    if (mirror && stage > e) e = stage;
    if (fused && stage + big_diag_scratch_elems() > e) e = stage + big_diag_scratch_elems();
    return e;
}

// The tile product, PIPELINED (round 4).  The round-3 kernel staged whole 64-deep k-blocks (67 KB of
// LDS: two workgroups per CU) and most of its launches are one to four rounds of workgroups, each a chain of memory
// latencies (operands -> LDS -> matrix cores -> C).  Here the k-dimension moves in chunks of 16 through two small
// buffers (37 KB: four workgroups per CU), one barrier per chunk, the next chunk's global loads in flight under the
// matrix instructions of the current one:
//   * staging: thread t loads 4 consecutive k of row t / 4 (one or two 128-bit loads; a wave covers 16 rows x 128 B)
//     and writes them as they are: As[row][k], row stride kGL = 18 doubles (20 floats);
//   * operands: matrix-instruction step s of a chunk takes k = 4 g + s from lane group g, so a lane's four A (B)
//     values of a chunk are CONSECUTIVE in LDS: one 128-bit read each for floats, two for doubles, on distinct banks
//     (36 c + 8 g dwords: sixteen different multiples of four mod 64);
//   * a mirrored tile goes out through LDS (rows of the transposed tile, coalesced) instead of 8-byte stores at a
//     stride of one matrix row.
template <class T, bool kFuse> QPX_DEV void big_gemm2_body(const Block& b, const BigGemmArgs<T>& a, int qp, int tile, T* lds)
{
    if (a.check_stop && a.ctrl && (a.ctrl + (size_t)qp * a.sctrl)[bcStop]) return;
    const int ti = tile / a.ntj, tj = tile - ti * a.ntj;
    const int crb = a.crb0 + ti, ccb = a.ccb0 + tj;
    if (a.lower && crb < ccb) return;
    constexpr int LD = big_gl<T>();
    T* As = lds;                         // [buf][row][LD]
    T* Bs = lds + 2 * kBB * LD;
    // operand blocks of k-block kb: A (arb0 + ti, akb0 + kb); B (brb0 + tj, bkb0 + kb), or (bkb0 + kb, brb0 + tj) when given as [k][column]
    const T* Ag = a.A + (size_t)qp * a.sA + big_blk(a.lda, a.arb0 + ti, a.akb0);
    const T* Bg = a.Bm + (size_t)qp * a.sB + (a.transb ? big_blk(a.ldb, a.bkb0, a.brb0 + tj) : big_blk(a.ldb, a.brb0 + tj, a.bkb0));
    const size_t bstep = a.transb ? (size_t)(a.ldb / kBB) * kBE : (size_t)kBE;        // from one k-block of B to the next
    const int lane = b.lane(), w = b.uniform(b.wave()), g = lane >> 4, c16 = lane & 15;
    const int qr = (w >> 1) * 32, qc = (w & 1) * 32;         // this wave's 32 x 32 quadrant
    T acc[2][2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[x][y][r] = T(0);
    T* Cq = a.C + (size_t)qp * a.sC;
    T* C = Cq + big_blk(a.ldc, crb, ccb);                                      // this tile: 4096 consecutive elements
    const T* Cs = a.Cs ? a.Cs + (size_t)qp * a.sCs + big_blk(a.ldcs, crb, ccb) : C;
    T cs[2][2][4];
    auto load_c = [&]() {
        if (!a.zero_init) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cs[x][y][r] = Cs[(qr + 16 * x + Block::mfma_row(T(0), g, r)) * kBB + qc + 16 * y + c16];
        }
    };
    load_c();
    // staging coordinates: A (and B as [column][k]): row sr, k 4 sq ..; B as [k][column]: k-row tr, columns 4 tq ..
    const int sr = b.tid >> 2, sq = b.tid & 3, tr = b.tid >> 4, tq = b.tid & 15;
    const int nch = a.nk * (kBB / kGC);
    // (one chunk of operands in flight per thread.  Two measured 5 % SLOWER at C4 with the C tile still fetched up front
    // -- four spilled registers at the 128-register limit of four workgroups per CU -- and EQUAL with the C tile fetched
    // after the k-loop instead (114 registers, no spill): the launches are not waiting for their operands.  profiles/archive/r04j, r04k.)
    T pa[4], pb[4];
    auto fetch = [&](int ch) {
        const int kb = ch >> 2, ko = (ch & 3) * kGC;                           // k-block, offset of the chunk inside it
        ld4(Ag + (size_t)kb * kBE + sr * kBB + ko + 4 * sq, pa);
        if (a.transb) ld4(Bg + (size_t)kb * bstep + (ko + tr) * kBB + 4 * tq, pb);
        else ld4(Bg + (size_t)kb * bstep + sr * kBB + ko + 4 * sq, pb);
    };
    fetch(0);
    for (int ch = 0; ch < nch; ++ch) {
        T* Ab = As + (ch & 1) * kBB * LD;
        T* Bb = Bs + (ch & 1) * kBB * LD;
        st4(Ab + sr * LD + 4 * sq, pa);
        if (a.transb) {
#pragma unroll
            for (int u = 0; u < 4; ++u) Bb[(4 * tq + u) * LD + tr] = pb[u];
        } else {
            st4(Bb + sr * LD + 4 * sq, pb);
        }
        b.sync();        // one barrier per chunk: the buffer written now was last read two chunks ago, before the previous barrier
        if (ch + 1 < nch) fetch(ch + 1);
        T a0[4], a1[4], b0[4], b1[4];
        ld4(Ab + (qr + c16) * LD + 4 * g, a0);
        ld4(Ab + (qr + 16 + c16) * LD + 4 * g, a1);
        ld4(Bb + (qc + c16) * LD + 4 * g, b0);
        ld4(Bb + (qc + 16 + c16) * LD + 4 * g, b1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            b.mfma16x16x4(a0[s], b0[s], acc[0][0]);
            b.mfma16x16x4(a0[s], b1[s], acc[0][1]);
            b.mfma16x16x4(a1[s], b0[s], acc[1][0]);
            b.mfma16x16x4(a1[s], b1[s], acc[1][1]);
        }
    }
    const bool fused = kFuse && a.fuse && tile == 0; // uniform
    const bool mir = a.mirror && crb != ccb;         // uniform
    const bool staged = fused || mir;
    if (staged) b.sync();                            // every wave is done with the operand buffers
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = qr + 16 * x + Block::mfma_row(T(0), g, r), jl = qc + 16 * y + c16;
                const int i = crb * kBB + il, j = ccb * kBB + jl;
                T v = a.zero_init ? T(0) : cs[x][y][r];
                if (a.dg && i == j) v += a.dg[(size_t)qp * a.sdg + i];
                v = fma_(a.alpha, acc[x][y][r], v);
                C[il * kBB + jl] = v;
                if (staged) lds[il * kBL + jl] = v;
            }
    if (!staged) return;
    b.sync();
    if (mir) {
        // row r of the transposed tile = column r of the staged one
        T* Cm = Cq + big_blk(a.ldc, ccb, crb);
        for (int r = w; r < kBB; r += b.nwaves()) Cm[r * kBB + lane] = lds[lane * kBL + r];
    }
    if constexpr (kFuse) {
        if (fused) {
            int* ctrl = a.ctrl ? a.ctrl + (size_t)qp * a.sctrl : nullptr;
            big_diag_block<T>(b, [&](int i, int j) { return lds[i * kBL + j]; },
                              a.W + (size_t)qp * a.sW + (size_t)a.fuse_k * 2 * kBB * kBB, ctrl, a.fail_bit, lds, lds + kBB * kBL, a.tile);
        }
    }
}

// ------------------------------------------------------------------------------------------ triangular solve
// x <- L^-1 xin (dir 0) or x <- L^-T xin (dir 1) for one QP per workgroup of sixteen waves, by blocks of 64.
// (round 5) BOTH directions read the LOWER triangle of the factor and the one copy W_kk of a diagonal block's inverse:
// the copy of L^T that rounds 2-4 kept above the diagonal (for the forward direction's access pattern) doubled the
// factor's footprint -- 2 MB + 0.13 MB per QP at C4, 270 MB for the batch against an Infinity Cache of 256 MB, read four
// times per pass by the two solves -- and cost 115 MB of mirrored writes per factorisation.  Wave w owns rows
// w, w + 16, w + 32, w + 48 of every 64 x 64 block; one load instruction is one 512-byte row segment.
//   forward  (block row k, blocks (k, j), j < k):  the wave's rows are OUTPUT rows -- a row dot with x_j, summed over
//            the lanes (DPP butterfly, no LDS); the diagonal block the same way with the rows of W_kk;
//   backward (block column k, blocks (j, k), j > k): the wave's rows are KNOWN entries -- every lane accumulates its
//            column, the sixteen partial sums meet in LDS; the diagonal block likewise with the rows of W_kk (= columns
//            of W_kk^T).
// The matrix entries of a block step do not depend on the solution: the first two off-diagonal blocks of step k + 1 and
// its rows of W (twelve loads per lane) are issued BEFORE step k's arithmetic and fly under its chain of reductions and
// barriers -- the round-4 kernel started every step with a cold round trip to HBM (~2 of its ~5 us per step); the rest
// of a step is issued at once when the step starts.  (All of a step one step ahead -- 2 x 32 doubles per lane -- does not
// fit the 128 registers of a sixteen-wave workgroup.)  The barriers of this kernel therefore wait for LDS traffic only
// (Block::sync_lds): the waves exchange nothing through global memory.
template <class T> struct BigTrsvArgs {
    int B, nb, dir, post;
    const T* M; size_t sM; int ld;
    const T* W; size_t sW;
    const T* xin; size_t sxin;            // right-hand side (may be the same array as x)
    T* x; size_t sx;
    const int* ctrl; size_t sctrl; int check_stop;
};
constexpr int kTrsvNW = 16;                    // waves per QP
constexpr int kTrsvRows = kBB / kTrsvNW;       // rows of a block a wave owns
constexpr int kTrsvPrefetch = 2;               // off-diagonal blocks of a step fetched one step ahead
QPX_LAYOUT_HD size_t big_trsv_lds_elems(int np) { return (size_t)np + (size_t)(kTrsvNW + 1) * kBB; }

// kLong (r6): more than eight blocks of 64 (beyond 512 unknowns): a step's streamed blocks come in rounds of MS.  A separate
// instantiation: with the rounds in the kernel the sizes up to 512 paid 1.5 - 14 % (registers of a sixteen-wave workgroup).
template <class T, bool kLong = false> QPX_DEV void big_trsv_body(const Block& b, const BigTrsvArgs<T>& a, int qp, T* lds)
{
    if (a.check_stop && a.ctrl && (a.ctrl + (size_t)qp * a.sctrl)[bcStop]) return;
    constexpr int NW = kTrsvNW, RW = kTrsvRows, PF = kTrsvPrefetch, MS = 512 / kBB - 1 - kTrsvPrefetch;      // (MS: streamed blocks of a step at the largest size)
    const int np = a.nb * kBB;
    T* xs = lds;                 // the vector
    T* part = xs + np;           // NW x 64 partial sums (backward direction)
    T* t = part + NW * kBB;      // block right-hand side
    const T* M = a.M + (size_t)qp * a.sM;
    const T* Wq = a.W + (size_t)qp * a.sW;
    T* x = a.x + (size_t)qp * a.sx;
    const T* xin = a.xin + (size_t)qp * a.sxin;
    const int lane = b.lane(), w = b.uniform(b.wave());
    const int dir = a.dir, nb = a.nb;
    // rows w + 16 u of off-diagonal block jb of step k (forward: block (k, jb); backward: block (k + 1 + jb, k)), element `lane`
    auto rows_of = [&](int k, int jb) {
        return M + (dir == 0 ? big_blk(a.ld, k, jb) : big_blk(a.ld, k + 1 + jb, k)) + (size_t)w * kBB + lane;
    };
    auto ld_rows = [&](const T* p, T (&v)[RW]) {
#pragma unroll
        for (int u = 0; u < RW; ++u) v[u] = p[(size_t)u * NW * kBB];
    };
    // what step k starts from: its first PF off-diagonal blocks and the rows of W_kk
    auto prefetch = [&](int k, T (&pm)[PF][RW], T (&pw)[RW]) {
        const int nblk = dir == 0 ? k : nb - 1 - k;
#pragma unroll
        for (int jb = 0; jb < PF; ++jb)
            if (jb < nblk) ld_rows(rows_of(k, jb), pm[jb]);
        ld_rows(Wq + (size_t)k * 2 * kBB * kBB + (size_t)w * kBB + lane, pw);
    };
    auto gather = [&](int i) {
        T sum = T(0);
#pragma unroll
        for (int ww = 0; ww < NW; ww += 4)
            sum += (part[ww * kBB + i] + part[(ww + 1) * kBB + i]) + (part[(ww + 2) * kBB + i] + part[(ww + 3) * kBB + i]);
        return sum;
    };
    T pm[PF][RW], pw[RW];
#pragma unroll
    for (int jb = 0; jb < PF; ++jb)
#pragma unroll
        for (int u = 0; u < RW; ++u) pm[jb][u] = T(0);
    prefetch(dir == 0 ? 0 : nb - 1, pm, pw);
    for (int i = b.tid; i < np; i += b.nt) xs[i] = xin[i];
    b.sync_lds();
    for (int kk = 0; kk < nb; ++kk) {
        const int k = dir == 0 ? kk : nb - 1 - kk;
        const int k0 = k * kBB;
        const int nblk = dir == 0 ? k : nb - 1 - k;
        T cm[PF][RW], cw[RW];
#pragma unroll
        for (int jb = 0; jb < PF; ++jb)
#pragma unroll
            for (int u = 0; u < RW; ++u) cm[jb][u] = pm[jb][u];
#pragma unroll
        for (int u = 0; u < RW; ++u) cw[u] = pw[u];
        if (kk + 1 < nb) prefetch(dir == 0 ? k + 1 : k - 1, pm, pw);
        // acc[u]: forward -- the dot of output row w + 16 u with the known x, one partial product per lane;
        //         backward -- the known rows w + 16 u times their x, accumulated per lane (= per output column)
        T acc[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) acc[u] = T(0);
        // x of off-diagonal block jb as this lane needs it: forward x_j[lane]; backward x_j[w + 16 u]
        auto mac = [&](int jb, const T (&mv)[RW]) {
            if (dir == 0) {
                const T xv = xs[jb * kBB + lane];
#pragma unroll
                for (int u = 0; u < RW; ++u) acc[u] = fma_(mv[u], xv, acc[u]);
            } else {
                const T* xj = xs + (k + 1 + jb) * kBB + w;
#pragma unroll
                for (int u = 0; u < RW; ++u) acc[u] = fma_(mv[u], xj[NW * u], acc[u]);
            }
        };
        // the blocks beyond the prefetched ones: ALL their loads at once (at most five blocks = twenty loads per lane at the
        // largest size) -- in pairs, the largest steps were three dependent round trips to HBM long
        if constexpr (!kLong) {
            T ms[MS][RW];
#pragma unroll
            for (int j = 0; j < MS; ++j)
                if (PF + j < nblk) ld_rows(rows_of(k, PF + j), ms[j]);
#pragma unroll
            for (int j = 0; j < MS; ++j)
                if (PF + j < nblk) mac(PF + j, ms[j]);
        } else {
            for (int j0 = PF; j0 < nblk; j0 += MS) {
                T ms[MS][RW];
#pragma unroll
                for (int j = 0; j < MS; ++j)
                    if (j0 + j < nblk) ld_rows(rows_of(k, j0 + j), ms[j]);
#pragma unroll
                for (int j = 0; j < MS; ++j)
                    if (j0 + j < nblk) mac(j0 + j, ms[j]);
            }
        }
#pragma unroll
        for (int jb = 0; jb < PF; ++jb)
            if (jb < nblk) mac(jb, cm[jb]);
        if (dir == 0) {
            // t_i = b_i - sum_j L(k, j)[i][:] . x_j for the wave's four rows i; then x_i = W_kk[i][:] . t
#pragma unroll
            for (int u = 0; u < RW; ++u) acc[u] = wave_sum(b, acc[u]);
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < RW; ++u) t[w + NW * u] = xs[k0 + w + NW * u] - acc[u];
            }
            b.sync_lds();
            const T tv = t[lane];
#pragma unroll
            for (int u = 0; u < RW; ++u) acc[u] = wave_sum(b, cw[u] * tv);
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < RW; ++u) xs[k0 + w + NW * u] = acc[u];
            }
            b.sync_lds();
        } else {
            // t_i = b_i - sum_j sum_c L(j, k)[c][i] x_j[c] over the wave's rows c; then x_i = sum_c W_kk[c][i] t[c]
            part[w * kBB + lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            b.sync_lds();
            if (b.tid < kBB) t[b.tid] = xs[k0 + b.tid] - gather(b.tid);
            b.sync_lds();
            T s2 = T(0);
#pragma unroll
            for (int u = 0; u < RW; ++u) s2 = fma_(cw[u], t[w + NW * u], s2);
            part[w * kBB + lane] = s2;
            b.sync_lds();
            if (b.tid < kBB) xs[k0 + b.tid] = gather(b.tid);
            b.sync_lds();
        }
    }
    for (int i = b.tid; i < np; i += b.nt) x[i] = a.post ? -xs[i] : xs[i];
}

// ------------------------------------------------------------------------------------------ mat-vec
// y[i] = beta y0[i] + alpha sum_j Mt[i][j] x[j]   (trans 0: rows of M; trans 1: y[j] = ... sum_i M[i][j] x[i])
// grid (B, ceil(out / 64)): 64 outputs per workgroup.  rows/cols are the logical extent that is read.
template <class T> struct BigGemvArgs {
    int B, rows, cols, trans;
    const T* M; size_t sM; int ld;
    const T* x; size_t sx;
    const T* y0; size_t sy0;
    T* y; size_t sy;
    T alpha, beta;
    const int* ctrl; size_t sctrl; int check_stop;
};
// LDS: the vector x (up to 1 024 elements) + 4 x 64 partial sums
// (sized by the vector there is: eight blocks up to 512 -- the measured footprint --, sixteen beyond)
QPX_LAYOUT_HD int big_gemv_x_blocks(int nx) { return nx <= 8 * kBB ? 8 : 16; }
QPX_LAYOUT_HD size_t big_gemv_lds_elems(int nx) { return (size_t)big_gemv_x_blocks(nx) * kBB + 4 * kWave; }
#ifndef QPX_BIG_GEMV_ROWS
#define QPX_BIG_GEMV_ROWS 4          // rows a wave works on at once (x two column blocks: loads in flight per lane)
#endif
template <class T> QPX_DEV void big_gemv_body(const Block& b, const BigGemvArgs<T>& a, int qp, int chunk, T* lds)
{
    if (a.check_stop && a.ctrl && (a.ctrl + (size_t)qp * a.sctrl)[bcStop]) return;
    const T* M = a.M + (size_t)qp * a.sM;
    const T* x = a.x + (size_t)qp * a.sx;
    const T* y0 = a.y0 ? a.y0 + (size_t)qp * a.sy0 : nullptr;
    T* y = a.y + (size_t)qp * a.sy;
    const int lane = b.lane(), w = b.uniform(b.wave()), nw = b.nwaves();
    // (round 4) The first form walked a row with ONE matrix load in flight per lane (the compiler waits for each load
    // of the run-time column loop before the next: `s_waitcnt vmcnt(0)` per iteration) and fetched x from global memory
    // beside it -- 3.7 TB/s on a part whose plain streaming read reaches 5.7 (scripts/bw_probe.py).  Now x sits in LDS
    // and a wave keeps RB rows x 2 column blocks = eight independent loads in flight.
    T* xl = lds;                                 // x, padded with zeros to whole blocks of 64
    T* part = lds + big_gemv_x_blocks(a.trans ? a.rows : a.cols) * kBB;
    const int nx = a.trans ? a.rows : a.cols, nxp = (nx + kBB - 1) / kBB * kBB;
    for (int i = b.tid; i < nxp; i += b.nt) xl[i] = i < nx ? x[i] : T(0);
    b.sync();
    if (!a.trans) {
        constexpr int RB = QPX_BIG_GEMV_ROWS;
        const int ncb = nxp / kBB;
        for (int r0 = w * RB; r0 < kBB; r0 += nw * RB) {
            const int i0 = chunk * kBB + r0;
            if (i0 >= a.rows) break;
            // rows i0 .. i0 + RB - 1 of block row `chunk` (the padded rows of the blob exist: loads stay unconditional)
            const T* row = M + big_blk(a.ld, chunk, 0) + (size_t)r0 * kBB + lane;
            T acc[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) acc[u] = T(0);
            int cb = 0;
            for (; cb + 2 <= ncb; cb += 2) {
                T m0[RB], m1[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) { m0[u] = row[(size_t)cb * kBE + u * kBB]; m1[u] = row[(size_t)(cb + 1) * kBE + u * kBB]; }
                const T x0 = xl[cb * kBB + lane], x1 = xl[(cb + 1) * kBB + lane];
#pragma unroll
                for (int u = 0; u < RB; ++u) acc[u] = fma_(m1[u], x1, fma_(m0[u], x0, acc[u]));
            }
            if (cb < ncb) {
                const T x0 = xl[cb * kBB + lane];
#pragma unroll
                for (int u = 0; u < RB; ++u) acc[u] = fma_(row[(size_t)cb * kBE + u * kBB], x0, acc[u]);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) acc[u] = wave_sum(b, acc[u]);
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < RB; ++u)
                    if (i0 + u < a.rows) y[i0 + u] = fma_(a.alpha, acc[u], y0 ? a.beta * y0[i0 + u] : T(0));
            }
        }
    } else {
        // 64 output columns, the rows dealt over the four waves (eight in flight per lane), partial sums combined through LDS
        const int j = chunk * kBB + lane;
        T a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
        auto at = [&](int i) { return M[big_blk(a.ld, i >> 6, chunk) + (size_t)(i & 63) * kBB + lane]; };
        int i = w;
        for (; i + 7 * nw < nxp; i += 8 * nw) {
            const T m0 = at(i), m1 = at(i + nw), m2 = at(i + 2 * nw), m3 = at(i + 3 * nw);
            const T m4 = at(i + 4 * nw), m5 = at(i + 5 * nw), m6 = at(i + 6 * nw), m7 = at(i + 7 * nw);
            a0 = fma_(m0, xl[i], a0);
            a1 = fma_(m1, xl[i + nw], a1);
            a2 = fma_(m2, xl[i + 2 * nw], a2);
            a3 = fma_(m3, xl[i + 3 * nw], a3);
            a4 = fma_(m4, xl[i + 4 * nw], a4);
            a5 = fma_(m5, xl[i + 5 * nw], a5);
            a6 = fma_(m6, xl[i + 6 * nw], a6);
            a7 = fma_(m7, xl[i + 7 * nw], a7);
        }
        for (; i < nxp; i += nw) a0 = fma_(at(i), xl[i], a0);
        part[w * kWave + lane] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
        b.sync();
        if (w == 0 && j < a.cols) {
            T s = T(0);
            for (int ww = 0; ww < nw; ++ww) s += part[ww * kWave + lane];
            y[j] = fma_(a.alpha, s, y0 ? a.beta * y0[j] : T(0));
        }
    }
}

// ------------------------------------------------------------------------------------------ symmetric mat-vec
// y = R x for the symmetric R of the loop (R z' once per pass, R 1 once per forward) from its LOWER block triangle alone
// (round 5): the row-dot kernel above read all of R -- 2 MB per QP and pass at C4, 258 MB per launch, an eighth of the bytes
// a pass moves -- and was the only reader of the upper triangle, which the pre-factorisation therefore no longer writes
// (115 MB of mirrored tiles).  Workgroup (qp, I) owns block row I: every block (I, J), J <= I, is loaded once, row by
// row (wave w: rows 16 w .. 16 w + 15, one 512-byte segment per load) and used twice -- times x_J for the rows' dots
// (per-lane partial products accumulated over J, one lane reduction per row at the end) and, for J < I, times x_I for
// the block's contribution to y_J (per-lane column sums, the four waves' sums met in LDS), which goes to a per-(I, J) slot
// of the workspace.  A second, tiny launch adds, in a fixed order, y_J = rows_J + sum_{I > J} slot(I, J).
template <class T> struct BigSymvArgs {
    int B, rows;                          // logical order of R (the padded blocks are identity / zero: loads stay unconditional)
    int stage;                            // 0: block rows -> ws; 1: the sum -> y
    const T* M; size_t sM; int ld;
    const T* x; size_t sx;
    T* y; size_t sy;
    T* ws; size_t sws;                    // per QP: (nb + nb * nb) * 64 elements -- rows part, then slot (I, J) at (nb + I nb + J) * 64
    const int* ctrl; size_t sctrl; int check_stop;
};
QPX_LAYOUT_HD size_t big_symv_ws_elems(int np) { const size_t nb = np / kBB; return (nb + nb * nb) * kBB; }
// LDS: x_I and the current x_J (2 x 64) + 4 x 64 column partial sums
QPX_LAYOUT_HD size_t big_symv_lds_elems() { return (size_t)6 * kBB; }
template <class T> QPX_DEV void big_symv_body(const Block& b, const BigSymvArgs<T>& a, int qp, int I, T* lds)
{
    if (a.check_stop && a.ctrl && (a.ctrl + (size_t)qp * a.sctrl)[bcStop]) return;
    const int nb = a.ld / kBB;
    T* ws = a.ws + (size_t)qp * a.sws;
    const int lane = b.lane(), w = b.uniform(b.wave());
    if (a.stage == 1) {
        T* y = a.y + (size_t)qp * a.sy;
        for (int i = b.tid; i < a.rows; i += b.nt) {
            const int J = i >> 6, c = i & 63;
            T sum = ws[(size_t)J * kBB + c];
            for (int I2 = J + 1; I2 < nb; ++I2) sum += ws[((size_t)nb + (size_t)I2 * nb + J) * kBB + c];
            y[i] = sum;
        }
        return;
    }
    const T* M = a.M + (size_t)qp * a.sM;
    const T* x = a.x + (size_t)qp * a.sx;
    T* xI = lds;                 // x of this block row (the coefficients of the column sums)
    T* xJ = xI + kBB;            // x of the current block column
    T* part = xJ + kBB;          // 4 x 64
    constexpr int RW = 16;       // rows of a block a wave owns
    if (b.tid < kBB) { const int i = I * kBB + b.tid; xI[b.tid] = i < a.rows ? x[i] : T(0); }
    T racc[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) racc[u] = T(0);
    for (int J = 0; J <= I; ++J) {
        b.sync();                // xJ / part of the previous block are consumed
        if (b.tid < kBB) { const int i = J * kBB + b.tid; xJ[b.tid] = i < a.rows ? x[i] : T(0); }
        const T* blk = M + big_blk(a.ld, I, J) + (size_t)(RW * w) * kBB + lane;
        T v[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) v[u] = blk[(size_t)u * kBB];
        b.sync();
        const T xv = xJ[lane];
        T cacc = T(0);
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            racc[u] = fma_(v[u], xv, racc[u]);
            cacc = fma_(v[u], xI[RW * w + u], cacc);
        }
        if (J < I) {
            part[w * kBB + lane] = cacc;
            b.sync();
            if (b.tid < kBB)
                ws[((size_t)nb + (size_t)I * nb + J) * kBB + b.tid] = (part[b.tid] + part[kBB + b.tid]) + (part[2 * kBB + b.tid] + part[3 * kBB + b.tid]);
        }
    }
#pragma unroll
    for (int u = 0; u < RW; ++u) racc[u] = wave_sum(b, racc[u]);
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < RW; ++u) ws[(size_t)I * kBB + RW * w + u] = racc[u];
    }
}

// ------------------------------------------------------------------------------------------ loop phases
// The vector work of the PDIPM loop: one wave per QP on the blob's vectors (NS = MP / 64 register slots).  The
// mathematics and the control flow are those of ipm_loop_body (qpx_grid.h), phase by phase:
//   0  initialise the state (z = s = 1, tau = 1, ...), stage p and c's constant part
//   1  start point: x = -T^-1 c is in vX -> shifts, z, s, best := start, vA = z'
//   2  residuals with vB = R z', best iterate, stop decision, d = s/z, affine right-hand side
//   3  affine step in vX -> step length, sigma, corrector right-hand side
//   4  corrector step in vX -> step length, update z, s, tau, vA = z'
//   5  outputs lam, slack, iters, status, best_resid; vA = best z' for the recovery of zhat
template <class T> struct BigPhaseArgs {
    int B, n, m, phase, it, maxIter, notImprovedLim, stall_policy;
    T* fac; size_t fac_stride;
    const T *p, *h; long long sp, sh;
    T eps;
    T *lam, *slack, *best_resid, *trace;
    int *iters, *status;
    int q; const T* bq; long long sb;     // equality constraints: b (B, q)
    int split;                            // phase 2 without d = s/z (phase 7 has written it)
    int io32;                             // T = double: p, h, b, lam, slack, best_resid, trace are float32 arrays (QPX_F32_WIDE)
};

template <class T, int NS> QPX_DEV void big_phase_body(const Block& b, const BigPhaseArgs<T>& a, int qp)
{
    const BigLayout L = big_layout(a.n, a.m, a.q);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    T* sc = F + L.scal;
    int* ctrl = reinterpret_cast<int*>(F + L.ctrl);
    const int m = a.m, n = a.n, lane = b.lane();
    const int io32 = a.io32;
    const T mT = (T)m;
    T *vP = F + L.v(bvP), *vC = F + L.v(bvC), *vR1 = F + L.v(bvR1), *vZ = F + L.v(bvZ), *vS = F + L.v(bvS);
    T *vA = F + L.v(bvA), *vB = F + L.v(bvB), *vD = F + L.v(bvD), *vBZ = F + L.v(bvBZ), *vBS = F + L.v(bvBS);
    T *vRH = F + L.v(bvRH), *vX = F + L.v(bvX), *vDZA = F + L.v(bvDZA), *vDSA = F + L.v(bvDSA), *vRSC = F + L.v(bvRSC);
    T *vRZ = F + L.v(bvRZ), *vRS = F + L.v(bvRS), *vONE = F + L.v(bvONE);
    if (a.phase == 6) {                          // end of the pre-factorisation: its failure bits are the status
        if (lane == 0) a.status[qp] = ctrl[bcFail];
        return;
    }
    if (a.phase == 0) {
        const In<T> pg(a.p, (size_t)qp * a.sp, io32), hg(a.h, (size_t)qp * a.sh, io32);
        const In<T> bg(a.q > 0 ? a.bq : nullptr, (size_t)qp * a.sb, io32);
        for (int i = lane; i < L.VP; i += kWave) {
            vP[i] = (i < n) ? pg[i] : T(0);
            vC[i] = (i < m) ? hg[i] : T(0);
            vD[i] = T(1);                        // 1 on the pad for every later factorisation
            vONE[i] = (i < m) ? T(1) : T(0);
            vZ[i] = vS[i] = vRZ[i] = vRS[i] = T(1);
            vA[i] = vB[i] = vR1[i] = vBZ[i] = vBS[i] = vRH[i] = vX[i] = vDZA[i] = vDSA[i] = vRSC[i] = T(0);
            (F + L.v(bvU))[i] = (F + L.v(bvW))[i] = (F + L.v(bvY))[i] = T(0);      // pads stay zero: the mat-vecs write the logical extent only
            (F + L.v(bvBQ))[i] = (i < a.q) ? bg[i] : T(0);
            (F + L.v(bvTB))[i] = (F + L.v(bvT1))[i] = (F + L.v(bvNU))[i] = T(0);
        }
        if (lane == 0) {
            sc[bsTau] = T(1); sc[bsBtau] = T(1); sc[bsSigz] = T(0); sc[bsSigs] = T(0); sc[bsBres] = Lim<T>::inf();
            sc[bsFeasPrev] = T(0); sc[bsAlphaPrev] = T(0); sc[bsMu] = T(0); sc[bsSzdot] = T(0);
            ctrl[bcFail] &= (QPX_ST_Q_NOT_SPD | QPX_ST_A_RANK);       // a previous loop's breakdown does not count
            ctrl[bcStop] = (ctrl[bcFail] != 0) ? 1 : 0;
            ctrl[bcNnot] = 0; ctrl[bcFloor] = 0; ctrl[bcSt] = 0; ctrl[bcIters] = 0;
        }
        return;
    }
    if (a.phase == 5) {
        const T bres = sc[bsBres];
        int st = ctrl[bcSt];
        const int iters = ctrl[bcIters];
        const int fail = ctrl[bcFail];
        if (fail & (QPX_ST_Q_NOT_SPD | QPX_ST_A_RANK)) {
            const T nanv = Lim<T>::inf() - Lim<T>::inf();
            for (int i = lane; i < m; i += kWave) { put_(a.lam, io32, (size_t)qp * m + i, nanv); put_(a.slack, io32, (size_t)qp * m + i, nanv); vA[i] = nanv; }
            if (lane == 0) { a.iters[qp] = 0; put_(a.best_resid, io32, (size_t)qp, Lim<T>::inf()); a.status[qp] |= fail; }
            return;
        }
        if (iters >= a.maxIter && !(bres < a.eps)) st |= QPX_ST_MAXITER;
        if (!(bres <= T(1))) st |= QPX_ST_INACCURATE;
        const T bts = sc[bsBtau] * sc[bsSigz];
        for (int i = lane; i < L.VP; i += kWave) {
            if (i < m) {
                const T bz = vBZ[i];
                put_(a.lam, io32, (size_t)qp * m + i, bz);
                put_(a.slack, io32, (size_t)qp * m + i, vBS[i]);
                vA[i] = bz - bts;
            } else {
                vA[i] = T(0);
            }
        }
        if (lane == 0) { a.iters[qp] = iters; a.status[qp] |= st | fail; put_(a.best_resid, io32, (size_t)qp, bres); }
        return;
    }
    if (ctrl[bcStop]) return;
    // a loop factorisation broke down: keep the best iterate.  Phase 2 behind the factorisation (split: the order in which
    // R z' runs beside it) first lets the iterate it was given compete for best and takes its stop decision -- that iterate
    // is complete, only the factorisation FOR THE NEXT STEP failed -- and honours the breakdown only if the QP would have
    // gone on (a converged QP's extra factorisation meets slacks of 1e-30: its breakdown is not the QP's).
    const bool late_fail = a.phase == 2 && a.split;
    if (ctrl[bcFail] && !late_fail) {
        if (lane == 0) { ctrl[bcSt] |= QPX_ST_KKT_BREAKDOWN; ctrl[bcStop] = 1; }
        if (a.phase == 1) for (int i = lane; i < m; i += kWave) { vBZ[i] = T(1); vBS[i] = T(1); }
        return;
    }
    if (a.phase == 1) {
        T x[NS];
        ld_slots<NS>(b, x, vX, m, T(0));
        T mnz = Lim<T>::inf(), mns = Lim<T>::inf();
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) { mnz = min2_(mnz, x[k]); mns = min2_(mns, -x[k]); }
        }
        mnz = wave_min(b, mnz);
        mns = wave_min(b, mns);
        const T sigz = (mnz < T(0)) ? (T(1) - mnz) : T(0);
        const T sigs = (mns < T(0)) ? (T(1) - mns) : T(0);
        if (lane == 0) { sc[bsSigz] = sigz; sc[bsSigs] = sigs; }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) {
                const T zk = x[k] + sigz, sk = -x[k] + sigs;
                vZ[i] = zk; vS[i] = sk; vA[i] = x[k]; vBZ[i] = zk; vBS[i] = sk;
            }
        }
        return;
    }
    if (a.phase == 7) {
        // d = s/z and the reciprocals alone (what the factorisation needs): the residuals -- phase 2 with split = 1 --
        // follow the factorisation, so that R z' (a mat-vec over 2 MB per QP) runs beside it on a side stream
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) {
                const T zk = vZ[i], sk = vS[i];
                const T rzk = rcp_(zk);
                vRZ[i] = rzk;
                vRS[i] = rcp_(sk);
                vD[i] = sk * rzk;
            }
        }
        return;
    }
    if (a.phase == 2) {
        const int it = a.it;
        const T tsz = sc[bsTau] * sc[bsSigz];
        T pri2 = 0, szdot = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) {
                const T zk = vZ[i], sk = vS[i];
                const T rz = sk - vC[i] - vB[i];
                pri2 = fma_(rz, rz, pri2);
                szdot = fma_(sk, zk, szdot);
                vRH[i] = vC[i] + vB[i] + tsz * vR1[i];
                if (!a.split) {
                    const T rzk = rcp_(zk);
                    vRZ[i] = rzk;
                    vRS[i] = rcp_(sk);
                    vD[i] = sk * rzk;
                }
            }
        }
        pri2 = wave_sum(b, pri2);
        szdot = wave_sum(b, szdot);
        const T mu = abs_(szdot / mT);
        const T pri = sqrt_(pri2);
        const T dual = tsz * sc[bsGt1];
        const T feas = pri + dual, resid = feas + mT * mu;
        const T tau = sc[bsTau];
        T bres = sc[bsBres];
        int nnot = ctrl[bcNnot], floor_hit = ctrl[bcFloor];
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
        if (a.stall_policy == 2 && it >= 1 && feas > T(2) * (T(1) - sc[bsAlphaPrev]) * sc[bsFeasPrev]) floor_hit = 1;
        if ((a.stall_policy != 0 && nnot >= a.notImprovedLim) || bres < a.eps || mu > T(1e32)) stopf = 1;
        if (a.stall_policy == 2 && floor_hit && mT * mu < T(1e-2) * feas) stopf = 1;
        const bool bad = !finite_(resid);
        if (bad) stopf = 1;
        const int failed = late_fail ? ctrl[bcFail] : 0;
        const bool broke = failed && !stopf;
        if (broke) stopf = 1;
        b.wave_sync();
        if (lane == 0) {
            if (broke) ctrl[bcSt] |= QPX_ST_KKT_BREAKDOWN;
            else if (failed) ctrl[bcFail] = failed & ~QPX_ST_KKT_BREAKDOWN;     // stopped anyway: the factorisation was never needed
            sc[bsMu] = mu; sc[bsSzdot] = szdot;
            ctrl[bcIters] = it + 1;
            if (better) { sc[bsBres] = bres; sc[bsBtau] = tau; }
            sc[bsFeasPrev] = feas;
            ctrl[bcNnot] = nnot; ctrl[bcFloor] = floor_hit;
            if (bad) ctrl[bcSt] |= QPX_ST_NONFINITE;
            ctrl[bcStop] = stopf;
            if (a.trace) {
                const size_t tr = ((size_t)it * a.B + qp) * 3;
                put_(a.trace, io32, tr, pri); put_(a.trace, io32, tr + 1, dual); put_(a.trace, io32, tr + 2, mu);
            }
        }
        return;
    }
    if (a.phase == 3) {
        const T mu = sc[bsMu], szdot = sc[bsSzdot];
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
        al = min2_(al, al2);
        al = min2_(al, T(1));
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
        return;
    }
    if (a.phase == 4) {
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
        al = min2_(al, al3);
        al = T(0.999) * al;
        al = min2_(al, T(1));
        const T tau = (T(1) - al) * sc[bsTau];
        const T tsz = tau * sc[bsSigz];
        b.wave_sync();
        if (lane == 0) { sc[bsTau] = tau; sc[bsAlphaPrev] = al; }
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
        return;
    }
}

// ------------------------------------------------------------------------------------------ small vector kernels
// op 0: y = alpha x + beta y0 (elementwise, len; out32: y is a float32 array of the caller, T = double);
// op 1: y[0] = || x[0..len) ||_2 (one wave);
// op 2: start of a pre-factorisation: zero the QP's control words, vONE = 1 on [0, m), 0 on the pad; vPQ = 0 on
//       [0, q), 1 on the pad (the diagonal of S11 = Yt Yt^T there)
template <class T> struct BigVecArgs {
    int B, op, len, n, m;
    T* fac; size_t fac_stride;
    const T *x, *y0; size_t sx, sy0;
    T* y; size_t sy;
    T alpha, beta;
    int q, out32;
};
template <class T> QPX_DEV void big_vec_body(const Block& b, const BigVecArgs<T>& a, int qp)
{
    if (a.op == 0) {
        const T* x = a.x + (size_t)qp * a.sx;
        const T* y0 = a.y0 ? a.y0 + (size_t)qp * a.sy0 : nullptr;
        for (int i = b.tid; i < a.len; i += b.nt) put_(a.y, a.out32, (size_t)qp * a.sy + i, fma_(a.alpha, x[i], y0 ? a.beta * y0[i] : T(0)));
    } else if (a.op == 1) {
        const T* x = a.x + (size_t)qp * a.sx;
        if (b.wave() == 0) {
            T acc = T(0);
            for (int i = b.lane(); i < a.len; i += kWave) acc = fma_(x[i], x[i], acc);
            acc = wave_sum(b, acc);
            if (b.lane() == 0) (a.y + (size_t)qp * a.sy)[0] = sqrt_(acc);
        }
    } else {
        const BigLayout L = big_layout(a.n, a.m, a.q);
        T* F = a.fac + (size_t)qp * a.fac_stride;
        int* ctrl = reinterpret_cast<int*>(F + L.ctrl);
        if (b.tid < 16) ctrl[b.tid] = 0;
        for (int i = b.tid; i < L.VP; i += b.nt) {
            (F + L.v(bvONE))[i] = (i < a.m) ? T(1) : T(0);
            (F + L.v(bvPQ))[i] = (i >= a.q && i < L.QP) ? T(1) : T(0);
        }
    }
}

// KKT solve / backward set-up and epilogue on the blob's vectors (one workgroup per QP)
//   stage 0: vD = 1/d (d given, or clamp(lam)/clamp(slack) for backward), vRH <- rs/d - rz, vU <- rx (n, padded 0),
//            vBQ <- ry (q; backward: 0)
//   stage 1: outputs: dz = vX, ds = (-rs - dz)/d, dx = vW, dy = vNU; backward: dp, dh, db and the outer products
template <class T> struct BigKktArgs {
    int B, n, m, stage, backward;
    T* fac; size_t fac_stride;
    const T *d, *rx, *rs, *rz;
    const T *zhat, *lam, *slack, *dl_dz;
    T *dx, *ds, *dz, *dQ, *dp, *dG, *dh;
    int* status;
    int q; const T *ry, *nu; T *dy, *dA, *db;
    int io32;                             // T = double: every array but `fac` is float32 (QPX_F32_WIDE)
};
template <class T> QPX_DEV void big_kkt_body(const Block& b, const BigKktArgs<T>& a, int qp, int chunk)
{
    const BigLayout L = big_layout(a.n, a.m, a.q);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    int* ctrl = reinterpret_cast<int*>(F + L.ctrl);
    const int n = a.n, m = a.m, q = a.q, io32 = a.io32;
    T *vD = F + L.v(bvD), *vRH = F + L.v(bvRH), *vU = F + L.v(bvU), *vX = F + L.v(bvX), *vW = F + L.v(bvW);
    T *vBQ = F + L.v(bvBQ), *vNU = F + L.v(bvNU);
    const In<T> rsg(a.backward ? nullptr : a.rs, (size_t)qp * m, io32);
    if (a.stage == 0) {
        const In<T> rxg(a.backward ? a.dl_dz : a.rx, (size_t)qp * n, io32), rzg(a.backward ? nullptr : a.rz, (size_t)qp * m, io32);
        const In<T> ryg((!a.backward && q > 0) ? a.ry : nullptr, (size_t)qp * q, io32), dg(a.backward ? nullptr : a.d, (size_t)qp * m, io32);
        const In<T> lamg(a.backward ? a.lam : nullptr, (size_t)qp * m, io32), slg(a.backward ? a.slack : nullptr, (size_t)qp * m, io32);
        for (int i = b.tid; i < L.VP; i += b.nt) {
            T dinv = T(1), rhs = T(0);
            if (i < m) {
                T d;
                if (a.backward) {
                    const T l = lamg[i], sl = slg[i];
                    d = ((l < T(1e-8)) ? T(1e-8) : l) / ((sl < T(1e-8)) ? T(1e-8) : sl);       // qp.py:148
                } else {
                    d = dg[i];
                }
                dinv = T(1) / d;
                rhs = (rsg ? rsg[i] * dinv : T(0)) - (rzg ? rzg[i] : T(0));
            }
            vD[i] = dinv;
            vRH[i] = rhs;
            vU[i] = (i < n && rxg) ? rxg[i] : T(0);
            vW[i] = T(0);
            vBQ[i] = (i < q && ryg) ? ryg[i] : T(0);
            (F + L.v(bvTB))[i] = (F + L.v(bvT1))[i] = vNU[i] = T(0);
        }
        if (b.tid == 0) { ctrl[bcStop] = 0; ctrl[bcFail] &= (QPX_ST_Q_NOT_SPD | QPX_ST_A_RANK); }
        return;
    }
    // stage 1, grid (B, 1 + rows of the outer products / 16)
    const bool bad = (ctrl[bcFail] & QPX_ST_KKT_BREAKDOWN) != 0;
    if (chunk == 0) {
        if (b.tid == 0 && bad && a.status) a.status[qp] |= QPX_ST_KKT_BREAKDOWN;
        for (int i = b.tid; i < n; i += b.nt) {
            const T v = bad ? T(0) : vW[i];
            if (a.backward && a.dp) put_(a.dp, io32, (size_t)qp * n + i, v);
            if (a.dx) put_(a.dx, io32, (size_t)qp * n + i, v);
        }
        for (int i = b.tid; i < m; i += b.nt) {
            const T dzv = bad ? T(0) : vX[i];
            if (a.backward && a.dh) put_(a.dh, io32, (size_t)qp * m + i, -dzv);
            if (a.dz) put_(a.dz, io32, (size_t)qp * m + i, dzv);
            if (!a.backward && a.ds) put_(a.ds, io32, (size_t)qp * m + i, (-(rsg ? rsg[i] : T(0)) - dzv) * vD[i]);
        }
        for (int i = b.tid; i < q; i += b.nt) {
            const T dyv = bad ? T(0) : vNU[i];
            if (a.backward && a.db) put_(a.db, io32, (size_t)qp * q + i, -dyv);
            if (a.dy) put_(a.dy, io32, (size_t)qp * q + i, dyv);
        }
        return;
    }
    if (!a.backward) return;
    const In<T> zh(a.zhat, (size_t)qp * n, io32), lm(a.lam, (size_t)qp * m, io32), nug(q > 0 ? a.nu : nullptr, (size_t)qp * q, io32);
    const int r0 = (chunk - 1) * 16;
    if (a.dQ)
        for (int e = b.tid; e < 16 * n; e += b.nt) {
            const int r = r0 + e / n, c = e % n;
            if (r < n) put_(a.dQ, io32, ((size_t)qp * n + r) * n + c, bad ? T(0) : T(0.5) * (vW[r] * zh[c] + zh[r] * vW[c]));
        }
    if (a.dG)
        for (int e = b.tid; e < 16 * n; e += b.nt) {
            const int r = r0 + e / n, c = e % n;
            if (r < m) put_(a.dG, io32, ((size_t)qp * m + r) * n + c, bad ? T(0) : (vX[r] * zh[c] + lm[r] * vW[c]));
        }
    if (q > 0 && a.dA)
        for (int e = b.tid; e < 16 * n; e += b.nt) {
            const int r = r0 + e / n, c = e % n;
            if (r < q) put_(a.dA, io32, ((size_t)qp * q + r) * n + c, bad ? T(0) : (vNU[r] * zh[c] + nug[r] * vW[c]));
        }
}

// ------------------------------------------------------------------------------------------ fused launches
// Fewer, longer kernels on the loop's critical path (a pass of the loop was 31 launches, many of them ~8 us):
//   big_solve_body: forward AND backward substitution of one solve, then the wave-0 phase that consumes the result
//   big_diag_body:  the wave-0 phase that produces d = s/z (and the stop decision), then the elimination of the first
//                   diagonal block of T = R + diag(d)
template <class T> struct BigSolveArgs {
    BigTrsvArgs<T> t;            // dir / post are set by the body: forward, then backward with the sign of `negate`
    BigPhaseArgs<T> ph;
    int negate, post_phase;      // post_phase < 0: none
    int pre_phase;               // > 0: the wave-0 phase that writes the right-hand side (and the stop flag) first
};
template <class T, int NS> QPX_DEV void big_solve_body(const Block& b, const BigSolveArgs<T>& a, int qp, T* lds)
{
    if (a.pre_phase > 0) {
        if (b.uniform(b.wave()) == 0) {
            BigPhaseArgs<T> ph = a.ph;
            ph.phase = a.pre_phase;
            big_phase_body<T, NS>(b, ph, qp);
        }
        b.sync();                                    // right-hand side and stop flag are in global memory for every wave
    }
    BigTrsvArgs<T> t = a.t;
    t.dir = 0; t.post = 0;
    big_trsv_body<T, (NS > 8)>(b, t, qp, lds);
    b.sync();
    t.dir = 1; t.post = a.negate; t.xin = a.t.x; t.sxin = a.t.sx;
    big_trsv_body<T, (NS > 8)>(b, t, qp, lds);
    if (a.post_phase >= 0) {
        b.sync();                                    // the solution is in global memory for wave 0
        if (b.uniform(b.wave()) == 0) {
            BigPhaseArgs<T> ph = a.ph;
            ph.phase = a.post_phase;
            big_phase_body<T, NS>(b, ph, qp);
        }
    }
}

template <class T> struct BigDiagArgs {
    BigPanelArgs<T> p;
    BigPhaseArgs<T> ph;
    int pre_phase;               // < 0: none
};
template <class T, int NS> QPX_DEV void big_diag_body(const Block& b, const BigDiagArgs<T>& a, int qp, T* lds)
{
    if (a.pre_phase >= 0) {
        if (b.uniform(b.wave()) == 0) {
            BigPhaseArgs<T> ph = a.ph;
            ph.phase = a.pre_phase;
            big_phase_body<T, NS>(b, ph, qp);
        }
        b.sync();
    }
    big_panel_body<T>(b, a.p, qp, lds);
}

}  // namespace qpx
