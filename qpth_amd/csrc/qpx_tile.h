// qpx_tile.h -- the PDIPM loop's matrix operations on MATRIX-CORE TILES (f64, gfx950).
//
// The m x m work matrix T = R + diag(s/z) lives in registers as 16x16 tiles of the lower block
// triangle, each tile in the C/D layout of v_mfma_f64_16x16x4_f64: lane l = 16 g + c of the
// owning wave holds, in register r, element (g + 4 r, c) of the tile.  Two facts about that
// layout carry the design:
//
//   * register r of a tile is a B operand (4 x 16, lane (g, c) gives B[g][c]) for the k-slice of rows g + 4 r as it
//     stands, and -- scaled per row -- an A operand (16 x 4, lane (g, c) gives A[c][g]) of the transposed tile;
//   * stored to LDS register by register ([r][lane]) a tile is plain row-major.
//
// ldl_inv (T = L~ D L~^T, with W~ = L~^-1 built in place of the eliminated columns, as in qpx_grid.h) is therefore
// BLOCKED BY SIXTEEN COLUMNS -- one tile column per panel, two barriers per panel; the scheme is described at
// panel16() below.  (The round-1 form, four columns per panel with the 4 x 4 pivot block factored redundantly by every
// wave, is kept behind -DQPX_TILE_PANEL4 for same-box A/B: profiles/r02k .. r02p.)
//
// NW waves share a QP (NW = 1, 2 or 4).  Tile rows are dealt round-robin from the bottom: wave w
// owns rows I_p = NBL-1 - p NW - (w or NW-1-w, alternating), p = 0 .. NPOS-1 ("positions"), and keeps tile (I_p, J) in
// register slot slot(p, J) -- a static index for static (p, J); which row a position is, is a
// wave-uniform scalar (a compile-time constant when NW = 1).  The tile row Ip of the current
// panel is a run-time value (the panel code exists once, not NBL times), tests against it are scalar branches.
//
// The triangular mat-vecs of the solve (x = -W~^T D^-1 W~ r) and the symmetric mat-vec R z use
// vector FMAs on the same registers; sums along tile rows and down tile columns go through LDS
// partials that are added in a fixed order (deterministic results).
#pragma once
#include "qpx_grid.h"

// Sub-phase timers of one panel (-DQPX_PANEL_PROF, scripts/prof_panel.py): thread 0 of every QP adds the
// shader-clock cycles of publish / barrier / pivot block / operands / update to qpx_panel_prof[].
#ifdef QPX_PANEL_PROF
static __device__ unsigned long long qpx_panel_prof[8];   // one copy per translation unit; TU 9 reads its own
#define QPX_PP(i)                                                        \
    {                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      \
        const long long qpx_pp_n = clock64();                            \
        pacc[i] += qpx_pp_n - pacc[7];                                   \
        pacc[7] = qpx_pp_n;                                              \
    }
#else
#define QPX_PP(i)
#endif

namespace qpx {

#ifndef QPX_TILE_PANEL4
// row stride of the panel's X rows (17 mod 32 doubles) and size of the region the mat-vec partials share with the
// operand tiles of the factorisation
QPX_LAYOUT_HD constexpr int tile_xs(int nbl) { return ((16 * nbl - 17 + 31) / 32) * 32 + 17; }
QPX_LAYOUT_HD constexpr int tile_union(int nbl, int nw)
{
    const int npos = (nbl + nw - 1) / nw, a = nw * nbl * 64 + nw * 16 * npos * 17, b = nbl * 256;
    return a > b ? a : b;
}
QPX_LAYOUT_HD constexpr size_t tile_scratch_elems(int nbl, int nw)
{
    return (size_t)16 * tile_xs(nbl) + 2 * 16 * 18 + 2 + tile_union(nbl, nw) + 16 * (size_t)nbl;
}
#else
QPX_LAYOUT_HD constexpr size_t tile_scratch_elems(int nbl, int nw)
{
    const size_t mp = 16 * (size_t)nbl, npos = (size_t)(nbl + nw - 1) / nw;
    return 8 * mp + 32 + (size_t)nw * nbl * 64 + (size_t)nw * 16 * npos * 17 + mp;
}
#endif

template <int NBL, int NW> struct TileMat {
    using T = double;
    static constexpr int NPOS = (NBL + NW - 1) / NW, NT = 64 * NW, MP = 16 * NBL;
    static constexpr int psize(int p) { return NBL - p * NW; }            // tiles of position p at most (over the waves)
    static constexpr int rowof(int p, int w) { return NBL - 1 - p * NW - ((p & 1) ? NW - 1 - w : w); }
    // Positions 2k and 2k+1 share one run of slots: 2k fills it from the bottom (slot = base + J), 2k+1
    // from the top (base + size - 1 - J).  In snake order their tile counts add up to the same number
    // for every wave, so nothing is wasted (NBL = 7, NW = 4: 7 slots per wave instead of 7 + 3).
    static constexpr int pairsize(int k)
    {
        int best = 0;
        for (int w = 0; w < NW; ++w) {
            const int a = rowof(2 * k, w) + 1 > 0 ? rowof(2 * k, w) + 1 : 0;
            const int c = (2 * k + 1 < NPOS && rowof(2 * k + 1, w) + 1 > 0) ? rowof(2 * k + 1, w) + 1 : 0;
            best = a + c > best ? a + c : best;
        }
        return best;
    }
    static constexpr int pairbase(int k)
    {
        int b = 0;
        for (int i = 0; i < k; ++i) b += pairsize(i);
        return b;
    }
    static constexpr int slot(int p, int J)
    {
        return (p & 1) ? pairbase(p / 2) + pairsize(p / 2) - 1 - J : pairbase(p / 2) + J;
    }
    static constexpr int NSLOT = pairbase((NPOS + 1) / 2);
    static constexpr int minrow(int p) { return NBL - 1 - p * NW - (NW - 1); }   // smallest row any wave has at position p
    static constexpr int NROW = 16 * NPOS;                                 // matrix rows a wave owns (at most)
    struct Pos {
        int tid, lane, w, g, c;
        QPX_DEV explicit Pos(const Block& blk)
            : tid(blk.tid), lane(blk.lane()), w(NW == 1 ? 0 : blk.uniform(blk.wave())), g(blk.lane() >> 4),
              c(blk.lane() & 15)
        {
        }
        // tile row of position p (< 0: none).  Rows are dealt from the bottom in snake order (w, then
        // NW-1-w, ...) so that the tile counts of the waves stay close as the factorisation retires rows
        QPX_DEV int row(int p) const { return NBL - 1 - p * NW - ((p & 1) ? NW - 1 - w : w); }   // = rowof(p, w)
    };
    struct Regs { T e[NSLOT][4]; };
#ifdef QPX_TILE_PANEL4
    // scratch: X (2 x 4 x MP) | S (2 x 16) | part (NW x NBL x 64) | red (NW x NROW x 17) | yrow (MP)
    static constexpr int kX = 0, kS = 8 * MP, kPart = kS + 32, kRed = kPart + NW * NBL * 64, kRow = kRed + NW * NROW * 17;
#else
    // scratch: X (16 x XS: the 16 old rows of a panel) | S, W (16 x SS: pivot block, its inverse factor) | flag |
    // { part (NW x NBL x 64) | red (NW x NROW x 17) } or, during a factorisation, BT (NBL x 256: the operand
    // tiles) | yrow (MP).  XS = 17 mod 32 and SS = 18 keep both the row-wise and the transposed accesses
    // (lane stride XS resp. SS doubles) on distinct LDS banks.
    static constexpr int XS = tile_xs(NBL), SS = 18;
    static constexpr int kX = 0, kS = 16 * XS, kW = kS + 16 * SS, kFlag = kW + 16 * SS, kPart = kFlag + 2;
    static constexpr int kRed = kPart + NW * NBL * 64, kBT = kPart;
    static constexpr int kRow = kPart + tile_union(NBL, NW);
#endif
    QPX_LAYOUT_HD static size_t scratch_elems() { return (size_t)kRow + MP; }
    static QPX_DEV void sync(const Block& blk)
    {
        if (NW == 1) blk.wave_sync();
        else blk.sync();
    }
    static QPX_DEV const T* image(const T* F, const FacLayout& lay) { return F + lay.Rm; }

    // img: tile (I, J), J <= I, at [(I (I + 1) / 2 + J) * 256 + r * 64 + lane]
    static QPX_DEV void load(const Block& blk, const Pos& p, Regs& E, const T* img)
    {
        const GlobalRows<T> rows(img, NBL * (NBL + 1) / 2 * 256, p.lane);
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J <= I) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) E.e[slot(pp, J)][r] = rows.row((I * (I + 1) / 2 + J) * 4 + r);
                }
            }
        }
    }

    static QPX_DEV void add_diag(const Pos& p, Regs& E, const T* vd)
    {
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J != I) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p.c == p.g + 4 * r) E.e[slot(pp, J)][r] += vd[16 * J + p.c];
            }
        }
    }

    // Sums over the 16 columns of every tile row of this wave: acc[pp][r] of lane (g, c) is a partial of
    // matrix row 16 I_pp + g + 4 r.  Through LDS: one padded line of 17 per row, one lane adds a line.
    template <class F>
    static QPX_DEV void row_reduce(const Block& blk, const Pos& p, const T (&acc)[NPOS][4], T* scr, F&& emit)
    {
        T* red = scr + kRed + p.w * (NROW * 17);
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((pp * 4 + r) * 4 + p.g) * 17 + p.c] = acc[pp][r];
        blk.wave_sync();
#pragma unroll
        for (int o0 = 0; o0 < NROW; o0 += 64) {
            const int o = o0 + p.lane;                       // o = (pp * 4 + r) * 4 + g
            if (o < NROW) {
                const T* q = red + o * 17;
                const T s = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]))) +
                            (((q[8] + q[9]) + (q[10] + q[11])) + ((q[12] + q[13]) + (q[14] + q[15])));
                const int I = p.row(o >> 4);
                if (I >= 0) emit(16 * I + (o & 3) + 4 * ((o >> 2) & 3), s);
            }
        }
    }

    // out[j] = +-(base[j] + the column partials of every wave), fixed order
    template <bool kNeg>
    static QPX_DEV void gather_cols(const Block& blk, const T* part, const T* base, T* out)
    {
        for (int j = blk.tid; j < MP; j += NT) {
            const int J = j >> 4, cc = j & 15;
            T sum = base[j];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const T* pp = part + (size_t)(w * NBL + J) * 64 + cc;
                sum += (pp[0] + pp[16]) + (pp[32] + pp[48]);
            }
            out[j] = kNeg ? -sum : sum;
        }
    }

    // vout = S vin for the symmetric matrix in E (diagonal tiles hold both triangles)
    static QPX_DEV void symv(const Block& blk, const Pos& p, const Regs& E, const T* vin, T* vout, T* scr)
    {
        T* part = scr + kPart;
        T* yrow = scr + kRow;
        T acc[NPOS][4], u[NPOS][4];
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[pp][r] = T(0);
                u[pp][r] = I >= 0 ? vin[16 * I + p.g + 4 * r] : T(0);
            }
        }
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            const T xj = vin[16 * J + p.c];
            T col = 0;
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                if (J >= psize(pp)) continue;
                const int I = p.row(pp);
                if (J > I) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[pp][r] = fma_(E.e[slot(pp, J)][r], xj, acc[pp][r]);
                if (J < I) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) col = fma_(E.e[slot(pp, J)][r], u[pp][r], col);
                }
            }
            part[(size_t)(p.w * NBL + J) * 64 + p.lane] = col;
        }
        row_reduce(blk, p, acc, scr, [&](int i, T s) { yrow[i] = s; });
        sync(blk);
        gather_cols<false>(blk, part, yrow, vout);
        sync(blk);
    }

#ifdef QPX_TILE_PANEL4
    // ---- one panel of ldl_inv: rows/columns k0 .. k0+3, k0 = 16 Ip + 4 SP.  gm[k] = (g == k) as 0/1.
    template <int SP>
    static QPX_DEV int panel(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int Ip, const T (&gm)[4],
                             int npos, int nend, long long (&pacc)[8])
    {
        QPX_PP(5)
        const int k0 = 16 * Ip + 4 * SP;
        T* X = scr + kX + (SP & 1) * 4 * MP;
        T* S = scr + kS + (SP & 1) * 16;
        const bool inpan = (p.c >> 2) == SP;
        const int kc = p.c & 3;
        // -- publish: the panel's four rows left of the panel (in the panel's own tile only the columns
        // left of it: the identity and the columns below are written by other lanes further down, and two
        // lanes must not write one address even if a GPU wave would order them) ...
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            if (p.row(pp) != Ip) continue;
#pragma unroll
            for (int J = 0; J < psize(pp); ++J)
                if (J < Ip || (J == Ip && p.c < 4 * SP)) X[p.g * MP + 16 * J + p.c] = E.e[slot(pp, J)][SP];
        }
        // ... the pivot block (identity in X, the block itself in S) and the four columns below it.
        // Row g + 4 r of tile row I lies below the panel iff I > Ip or r > SP.  The panel's own columns
        // restart from zero: they are published, and the update writes -l~ W into them.
        if (inpan) {
#pragma unroll
            for (int J = 0; J < NBL; ++J) {
                if (J != Ip) continue;
#pragma unroll
                for (int pp = 0; pp < NPOS; ++pp) {
                    if (J >= psize(pp)) continue;
                    const int I = p.row(pp);
                    if (I < Ip) continue;
                    if (I == Ip) {
                        S[p.g * 4 + kc] = E.e[slot(pp, J)][SP];
                        X[p.g * MP + 16 * J + p.c] = (kc == p.g) ? T(1) : T(0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (I > Ip || r > SP) X[kc * MP + 16 * I + p.g + 4 * r] = E.e[slot(pp, J)][r];
                        E.e[slot(pp, J)][r] = T(0);
                    }
                }
            }
        }
        QPX_PP(0)
        sync(blk);
        QPX_PP(1)
        // -- the 4 x 4 pivot block: S = L D L^T, W = L^-1 (unit lower), every lane the same numbers
        const T s00 = S[0], s10 = S[4], s11 = S[5], s20 = S[8], s21 = S[9], s22 = S[10];
        const T s30 = S[12], s31 = S[13], s32 = S[14], s33 = S[15];
        const T d0 = s00, r0 = rcp_(d0);
        const T l10 = s10 * r0, l20 = s20 * r0, l30 = s30 * r0;
        const T d1 = fma_(-l10, s10, s11), r1 = rcp_(d1);
        const T v21 = fma_(-l20, s10, s21), v31 = fma_(-l30, s10, s31);
        const T l21 = v21 * r1, l31 = v31 * r1;
        const T d2 = fma_(-l21, v21, fma_(-l20, s20, s22)), r2 = rcp_(d2);
        const T v32 = fma_(-l31, v21, fma_(-l30, s20, s32));
        const T l32 = v32 * r2;
        const T d3 = fma_(-l32, v32, fma_(-l31, v31, fma_(-l30, s30, s33))), r3 = rcp_(d3);
        // pivots k < npos must be positive, npos <= k < nend negative (the equality block of the
        // augmented matrix in the pre-factorisation), the rest (identity padding) positive
        const T big = T(1e300);                  // NaN fails d > 0, +/-inf fails |d| < big
        const T g0 = (k0 >= npos && k0 < nend) ? -d0 : d0, g1 = (k0 + 1 >= npos && k0 + 1 < nend) ? -d1 : d1;
        const T g2 = (k0 + 2 >= npos && k0 + 2 < nend) ? -d2 : d2, g3 = (k0 + 3 >= npos && k0 + 3 < nend) ? -d3 : d3;
        const bool good = (g0 > T(0)) && (g1 > T(0)) && (g2 > T(0)) && (g3 > T(0)) && (g0 < big) && (g1 < big) &&
                          (g2 < big) && (g3 < big);
        if (!good) {
            const bool posok = !(k0 < npos && !(g0 > T(0) && g0 < big)) && !(k0 + 1 < npos && !(g1 > T(0) && g1 < big)) &&
                               !(k0 + 2 < npos && !(g2 > T(0) && g2 < big)) && !(k0 + 3 < npos && !(g3 > T(0) && g3 < big));
            return posok ? 2 : 1;               // 1: a pivot that must be positive is not; 2: one that must be negative
        }
        const T w10 = -l10, w21 = -l21, w32 = -l32;
        const T w20 = fma_(l21, l10, -l20);
        const T w31 = fma_(l32, l21, -l31);
        const T w30 = fma_(-w32, l20, fma_(-w31, l10, -l30));
        // row g of W and 1/d_g of this lane's group: sums against the 0/1 group masks (branch-free,
        // and cheaper than chains of 64-bit selects)
        const T cg0 = fma_(gm[3], w30, fma_(gm[2], w20, gm[1] * w10));
        const T cg1 = fma_(gm[3], w31, gm[2] * w21);
        const T cg2 = gm[3] * w32;
        const T rg = fma_(gm[3], r3, fma_(gm[2], r2, fma_(gm[1], r1, gm[0] * r0)));
        if (p.w == 0 && p.c == 0) rd[k0 + p.g] = rg;
        QPX_PP(2)
        // -- operands: bop[J] is the B operand of tile column J (the new W~ rows of the panel left of it,
        // L~_pp^-1 inside it, the un-scaled columns right of it); the A operand of tile row I is the same
        // combination at position 16 I + c, times -1/d_g, with zeros for rows not below the panel
        const T* Xg = X + p.g * MP;
        T bop[NBL];
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            const int j = 16 * J + p.c;
            bop[J] = fma_(cg2, X[2 * MP + j], fma_(cg1, X[MP + j], fma_(cg0, X[j], Xg[j])));
        }
        const T nrg = -rg;
        T aop[NPOS];
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
            aop[pp] = T(0);
            if (I < Ip) continue;
            const int i = 16 * I + p.c;
            const T t = fma_(cg2, X[2 * MP + i], fma_(cg1, X[MP + i], fma_(cg0, X[i], Xg[i])));
            aop[pp] = (I > Ip || p.c > 4 * SP + 3) ? t * nrg : T(0);
        }
        QPX_PP(3)
        // -- the panel's own rows are final: W~ rows left of the panel, L~_pp^-1 inside it.  (The diagonal
        // entries get the 1 of the unit factor; the d_k live in rd[] as reciprocals, E's diagonal is never
        // read after the factorisation.)
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            if (p.row(pp) != Ip) continue;
#pragma unroll
            for (int J = 0; J < psize(pp); ++J)
                if (J <= Ip) E.e[slot(pp, J)][SP] = bop[J];
        }
        // -- rank-4 update of every owned tile at or below the panel's tile row
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
            if (I < Ip) continue;
#pragma unroll
            for (int J = 0; J < psize(pp); ++J)
                if (J <= minrow(pp) || J <= I) blk.mfma16x16x4(aop[pp], bop[J], E.e[slot(pp, J)]);
        }
        QPX_PP(4)
        return 0;
    }

    // E: symmetric matrix (padded with the identity); eliminates columns 0 .. ncol-1 (rounded up to a
    // panel): strictly lower part of those columns -> W~ = L~^-1 (rows beyond ncol: -(row block) W~),
    // trailing block -> Schur complement, rd[k] = 1/d_k.  Pivots k < npos must be positive, npos <= k <
    // nend negative, others positive.  Uniform return value: 0, or 1 / 2 = a positive / negative pivot
    // broke down.
    static QPX_DEV int ldl_inv_signed(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int ncol, int npos, int nend)
    {
        const T gm[4] = {p.g == 0 ? T(1) : T(0), p.g == 1 ? T(1) : T(0), p.g == 2 ? T(1) : T(0), p.g == 3 ? T(1) : T(0)};
        int fail = 0;
        long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef QPX_PANEL_PROF
        pacc[7] = clock64();
#endif
#pragma unroll 1
        for (int Ip = 0; Ip < NBL && !fail; ++Ip) {
            const int k0 = 16 * Ip;
            if (k0 >= ncol) break;
            fail = panel<0>(blk, p, E, scr, rd, Ip, gm, npos, nend, pacc);
            if (!fail && k0 + 4 < ncol) fail = panel<1>(blk, p, E, scr, rd, Ip, gm, npos, nend, pacc);
            if (!fail && k0 + 8 < ncol) fail = panel<2>(blk, p, E, scr, rd, Ip, gm, npos, nend, pacc);
            if (!fail && k0 + 12 < ncol) fail = panel<3>(blk, p, E, scr, rd, Ip, gm, npos, nend, pacc);
        }
#ifdef QPX_PANEL_PROF
        if (p.tid == 0) {
            for (int i = 0; i < 6; ++i) atomicAdd(&qpx_panel_prof[i], (unsigned long long)pacc[i]);
            atomicAdd(&qpx_panel_prof[6], 1ull);
        }
#endif
        sync(blk);
        return fail;
    }
    // E: T (SPD, order m, padded with the identity) -> strictly lower: W~ = L~^-1, rd[k] = 1/d_k.
    static QPX_DEV bool ldl_inv(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int m)
    {
        return ldl_inv_signed(blk, p, E, scr, rd, m, MP, MP) == 0;
    }

#else
    // ---- ldl_inv BLOCKED BY SIXTEEN COLUMNS (one tile column per panel, two barriers per panel).
    //
    // Panel Ip = rows/columns 16 Ip .. 16 Ip + 15, pivot block P = E(Ip, Ip) = L~_pp D L~_pp^T, W_pp = L~_pp^-1:
    //   publish   X_J (16 x 16, J = 0 .. NBL-1, J != Ip) = the panel's sixteen "old" rows: the W~ entries E(Ip, J) left
    //             of the panel (from the wave that owns tile row Ip), the panel's columns read down the matrix,
    //             E(J, Ip)^T, right of it (from the owners of those tiles; the update restarts them from zero).
    //   factor    the owner of tile row Ip moves P to "lane (g, c) = row c, columns 4 g .. 4 g + 3" (through LDS, inside
    //             the wave) and eliminates it there: per pivot one v_rcp_f64_dpp + Newton, the multipliers copied to
    //             the four lane groups by lane swaps, and ONE v_fmac_f64_dpp per register -- the pivot row arrives
    //             through the DPP row broadcast; nothing is published, no barrier.  The same rank-1 update builds W~
    //             in place of the eliminated columns (as everywhere in this file).  It overlaps with the other
    //             waves' publish.                                                       -- barrier A --
    //   operands  b_J = W_pp X_J on the matrix core (4 MFMAs per J, the J dealt over the waves; b_Ip = W_pp), written
    //             to LDS in the accumulator layout.                                   -- barrier B --
    //   update    E(Ip, J) = b_J (the panel's own rows are final);  E(I, J) += (-D^-1 b_I)^T b_J for I > Ip, J <= I:
    //             register r of b_J is the B operand of k-slice r as it is, and register r of b_I times -1/d is the
    //             A operand (the accumulator layout indexes both by (row g + 4 r, column c)).
    // Per sixteen columns: 2 barriers (four-column panels: 4), ~30 + 16 x 25 (one wave) vector instructions per
    // wave around the MFMAs (four-column panels: 4 x 230 in every wave).
    // Pivot K of the 16 x 16 pivot block held as lane (g, c) = row c, columns 4 g .. 4 g + 3 (four registers) plus,
    // in every group, the row's own diagonal entry dg (updated by dg -= l~^2 d, which needs nothing from other
    // lanes): the pivot d_K is then lane K's dg in every group, so its reciprocal (the long chain: estimate + two
    // Newton steps) runs beside the lane swaps that copy column K from its group K / 4 to the other three.  Every
    // lane then updates its four columns with the pivot row from lane K of its own row of 16 lanes.
    template <int K>
    static QPX_DEV void pivot16(const Block& blk, const Pos& p, T (&a)[4], T& dg, T& myr)
    {
        constexpr int GK = K / 4, KK = K % 4;
        const T dk = blk.template row_bcast<K>(dg);
        const T r = rcp_(dk);
        myr = p.lane == K ? r : myr;                             // (the pivots are checked together, after the block)
        if constexpr (K < 15) {
            const T v = blk.template grp_bcast<GK>(p.c > K ? a[KK] : T(0));   // column K below the pivot, 0 above
            const T nl = -(v * r);                               // -l~
            blk.template row_rank1<K>(a, nl);
            a[KK] = p.g == GK ? nl : a[KK];                      // column K: assigned (see qpx_grid.h); 0 on and above the diagonal
            dg = fma_(nl, v, dg);
        } else {
            a[KK] = p.g == GK ? T(0) : a[KK];
        }
    }

    static QPX_DEV bool panel16(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int Ip, int m, long long (&pacc)[8])
    {
        QPX_PP(5)
        T* X = scr + kX;
        T* S = scr + kS;
        T* W = scr + kW;
        T* BT = scr + kBT;
        T* flag = scr + kFlag;
        const int k0 = 16 * Ip;
        const int kmax = m - k0;              // pivots of this block that are not identity padding (>= 16: all)
        bool mine = false;
        // -- publish
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
            if (I < Ip) continue;
            if (I == Ip) mine = true;
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J > Ip) continue;
                if (I == Ip) {
                    if (J < Ip) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[(p.g + 4 * r) * XS + 16 * J + p.c] = E.e[slot(pp, J)][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) S[(p.g + 4 * r) * SS + p.c] = E.e[slot(pp, J)][r];
                    }
                } else if (J == Ip) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[p.c * XS + 16 * I + p.g + 4 * r] = E.e[slot(pp, J)][r];
                }
            }
        }
        QPX_PP(0)
        // -- the pivot block, by the wave that owns it.  Pivots of the identity padding are skipped (d = 1, no
        // multipliers): their columns of W are zero.
        if (mine) {
            blk.wave_sync();
            T a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = S[p.c * SS + 4 * p.g + j];
            T dg = S[p.c * SS + p.c], myr = T(1);
            pivot16<0>(blk, p, a, dg, myr);
            if (kmax > 1) pivot16<1>(blk, p, a, dg, myr);
            if (kmax > 2) pivot16<2>(blk, p, a, dg, myr);
            if (kmax > 3) pivot16<3>(blk, p, a, dg, myr);
            if (kmax > 4) pivot16<4>(blk, p, a, dg, myr);
            if (kmax > 5) pivot16<5>(blk, p, a, dg, myr);
            if (kmax > 6) pivot16<6>(blk, p, a, dg, myr);
            if (kmax > 7) pivot16<7>(blk, p, a, dg, myr);
            if (kmax > 8) pivot16<8>(blk, p, a, dg, myr);
            if (kmax > 9) pivot16<9>(blk, p, a, dg, myr);
            if (kmax > 10) pivot16<10>(blk, p, a, dg, myr);
            if (kmax > 11) pivot16<11>(blk, p, a, dg, myr);
            if (kmax > 12) pivot16<12>(blk, p, a, dg, myr);
            if (kmax > 13) pivot16<13>(blk, p, a, dg, myr);
            if (kmax > 14) pivot16<14>(blk, p, a, dg, myr);
            if (kmax > 15) pivot16<15>(blk, p, a, dg, myr);
#pragma unroll
            for (int j = 0; j < 4; ++j) W[p.c * SS + 4 * p.g + j] = (4 * p.g + j < kmax) ? a[j] : T(0);
            if (p.lane < 16) rd[k0 + p.lane] = myr;
            // a pivot that is not positive and finite leaves a reciprocal that is not (negative, NaN from inf - inf or
            // 0 * inf further down, 0 or inf); nothing above traps, so one test of the sixteen reciprocals replaces
            // two compares in every pivot's chain
            const bool bad = blk.any(!(myr > T(0) && myr < T(1e300)));
            if (p.lane == 0) flag[0] = bad ? T(1) : T(0);
        }
        QPX_PP(1)
        sync(blk);
        QPX_PP(2)
        const T zr = flag[0];                 // 0 from here on
        if (zr != T(0)) return false;
        blk.template prio<0>();               // the matrix-instruction streams yield to the other wave's chains
        // -- operand tiles: b_J = (I + W_strict) X_J
        {
            T wa[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wa[s] = W[p.c * SS + p.g + 4 * s];
#pragma unroll
            for (int jj = 0; jj < (NBL + NW - 1) / NW; ++jj) {
                const int J = p.w + jj * NW;
                if (J >= NBL) continue;
                T acc[4];
                if (J == Ip) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = W[(p.g + 4 * r) * SS + p.c] + ((p.g + 4 * r == p.c) ? T(1) : T(0));
                } else {
                    T bx[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[s] = bx[s] = X[(p.g + 4 * s) * XS + 16 * J + p.c];
#pragma unroll
                    for (int s = 0; s < 4; ++s) blk.mfma16x16x4(wa[s], bx[s], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) BT[J * 256 + r * 64 + p.lane] = acc[r];
            }
        }
        QPX_PP(3)
        sync(blk);
        QPX_PP(4)
        // -- update
        T aop[NPOS][4];
        {
            T nrd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) nrd[r] = -rd[k0 + p.g + 4 * r];
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int r = 0; r < 4; ++r) aop[pp][r] = T(0);
                if (I <= Ip) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) aop[pp][r] = BT[I * 256 + r * 64 + p.lane] * nrd[r];
            }
        }
        // the panel's own rows are final: W~ rows left of the panel, L~_pp^-1 inside it -- read straight into the tile
        // registers (an assignment from registers another path also uses costs round-trip copies of the tile)
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            if (p.row(pp) != Ip) continue;
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J > Ip) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) E.e[slot(pp, J)][r] = BT[J * 256 + r * 64 + p.lane];
            }
        }
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            bool need = false;
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                if (J >= psize(pp)) continue;
                const int I = p.row(pp);
                need = need || (I > Ip && J <= I);
            }
            if (!need) continue;
            T bj[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bj[r] = BT[J * 256 + r * 64 + p.lane];
            // the panel's own columns restart from zero (their old values went into X): multiplied by a factor that is
            // 0 there and 1 elsewhere, outside every branch
            const T keep = J == Ip ? zr : T(1);
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                if (J >= psize(pp)) continue;
                const int I = p.row(pp);
                if (I > Ip && J <= I) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) E.e[slot(pp, J)][r] *= keep;
#pragma unroll
                    for (int s = 0; s < 4; ++s) blk.mfma16x16x4(aop[pp][s], bj[s], E.e[slot(pp, J)]);
                }
            }
        }
        blk.template prio<3>();
        return true;
    }

    // E: T (SPD, order m, padded with the identity) -> strictly lower: W~ = L~^-1, rd[k] = 1/d_k; false: a pivot
    // broke down (uniform)
    static QPX_DEV bool ldl_inv(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int m)
    {
        bool ok = true;
        blk.template prio<3>();
        long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef QPX_PANEL_PROF
        pacc[7] = clock64();
#endif
#pragma unroll 1
        for (int Ip = 0; Ip < NBL && ok && 16 * Ip < m; ++Ip) ok = panel16(blk, p, E, scr, rd, Ip, m, pacc);
#ifdef QPX_PANEL_PROF
        if (p.tid == 0) {
            for (int i = 0; i < 6; ++i) atomicAdd(&qpx_panel_prof[i], (unsigned long long)pacc[i]);
            atomicAdd(&qpx_panel_prof[6], 1ull);
        }
#endif
        sync(blk);
        return ok;
    }
#endif
    // vout = -T^-1 vin = -W~^T D^-1 W~ vin (W~ unit lower in E, strictly lower part stored)
    static QPX_DEV void solve_neg(const Block& blk, const Pos& p, const Regs& E, const T* rd, int m, const T* vin,
                                  T* vout, T* tmp, T* scr)
    {
        T* part = scr + kPart;
        // u = D^-1 W~ vin: sums along tile rows, complete inside the owning wave
        T acc[NPOS][4];
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[pp][r] = T(0);
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            const T xj = vin[16 * J + p.c];
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                if (J >= psize(pp)) continue;
                const int I = p.row(pp);
                if (J > I) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T e = (J < I || p.c < p.g + 4 * r) ? E.e[slot(pp, J)][r] : T(0);
                    acc[pp][r] = fma_(e, xj, acc[pp][r]);
                }
            }
        }
        row_reduce(blk, p, acc, scr, [&](int i, T s) { tmp[i] = (i < m) ? (s + vin[i]) * rd[i] : T(0); });
        blk.wave_sync();      // a wave owns whole tile rows: the u it reads next are the ones it just wrote
        // x = W~^T u: sums down columns, partial per wave, gathered in a fixed order
        T u[NPOS][4];
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
#pragma unroll
            for (int r = 0; r < 4; ++r) u[pp][r] = I >= 0 ? tmp[16 * I + p.g + 4 * r] : T(0);
        }
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            T col = 0;
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                if (J >= psize(pp)) continue;
                const int I = p.row(pp);
                if (J > I) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T e = (J < I || p.c < p.g + 4 * r) ? E.e[slot(pp, J)][r] : T(0);
                    col = fma_(e, u[pp][r], col);
                }
            }
            part[(size_t)(p.w * NBL + J) * 64 + p.lane] = col;
        }
        sync(blk);
        gather_cols<true>(blk, part, tmp, vout);
        sync(blk);
    }
};

QPX_LAYOUT_HD size_t lds_elems_ipm_tile(int nbl, int nw, int n, int q)
{
    return lds_elems_ipm_loop(16 * (size_t)nbl, tile_scratch_elems(nbl, nw), n, q);
}

QPX_LAYOUT_HD size_t lds_elems_kkt_tile(int nbl, int nw, int n, int q)
{
    return lds_elems_kkt_mat(16 * (size_t)nbl, tile_scratch_elems(nbl, nw), n, q);
}

template <int NBL, int NW, bool kBackward>
QPX_DEV void kkt_tile_body(const Block& b, const KktArgs<double>& a, int qp, double* lds)
{
    kkt_mat_body<double, TileMat<NBL, NW>, kBackward>(b, a, qp, lds);
}

template <int NBL, int NW, int NS>
QPX_DEV void ipm_tile_body(const Block& b, const IpmArgs<double>& a, int qp, double* lds)
{
    ipm_loop_body<double, TileMat<NBL, NW>, NS>(b, a, qp, lds);
}

}  // namespace qpx
