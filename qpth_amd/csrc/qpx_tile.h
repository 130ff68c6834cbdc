// qpx_tile.h -- the PDIPM loop's matrix operations on MATRIX-CORE TILES (f64, gfx950).
//
// The m x m work matrix T = R + diag(s/z) lives in registers as 16x16 tiles of the lower block
// triangle, each tile in the C/D layout of v_mfma_f64_16x16x4_f64: lane l = 16 g + c of the
// owning wave holds, in register r, element (g + 4 r, c) of the tile.  Two facts about that
// layout carry the design:
//
//   * a panel of four consecutive rows 4 sp .. 4 sp + 3 of a tile is register `sp` of all 64
//     lanes (row 4 sp + g in lane group g), and a panel of four columns is 16 lanes;
//   * the B operand (4 x 16, lane (g, c) gives B[g][c]) and the A operand (16 x 4, lane (g, c)
//     gives A[c][g]) of the rank-4 update are both indexed by (panel column g, position c).
//
// ldl_inv (T = L~ D L~^T, with W~ = L~^-1 built in place of the eliminated columns, as in
// qpx_grid.h) is therefore BLOCKED BY FOUR COLUMNS.  Per panel, one LDS publish + one barrier:
//
//   publish   X[kk][j], j = 0 .. MP-1: the four "old" panel rows -- W~ entries (j < k0) from the
//             owner of the panel's tile row, the identity for the panel's own columns, and the
//             panel's columns read down the matrix (j > k0 + 3) from the owners of those rows;
//             S = the 4 x 4 pivot block.
//   every wave factors S redundantly in registers (4 reciprocals, ~50 flops; the result is
//             uniform, so is the breakdown decision) and forms, per 16 columns J,
//             b_J[g][c] = sum_{kk' <= g} (L~_pp^-1)[g][kk'] X[kk'][16 J + c]
//             which is at once: the new W~ rows of the panel (J left of the panel), L~_pp^-1 itself
//             (the panel's columns), the un-scaled columns v (J right of it), and -- scaled by
//             -1/d_g -- the A operand of tile row J.
//   update    one v_mfma_f64_16x16x4 per owned tile below the panel: E += (-l~) b.
//
// That is 1 matrix instruction per tile and 4 columns where the thread-grid kernel issues
// 4 x 28 vector FMAs per thread, and the ~60 instructions of per-column bookkeeping are paid once
// per four columns.  Tile rows are dealt to waves in pairs (I, NBL-1-I or so) so that every wave
// owns NBL (or NBL+1) tiles: tile (I, J) sits in slot J of the wave that has I as its "row A" and
// in slot NSLOT-1-J of the wave that has it as its "row B" -- all register indices are static once
// the panel index is a template parameter, ownership tests are wave-uniform branches.
//
// The triangular mat-vecs of the solve (x = -W~^T D^-1 W~ r) and the symmetric mat-vec R z use
// vector FMAs on the same registers; sums along a tile row are DPP butterflies inside 16-lane
// rows (no LDS), sums down a column go through LDS partials in a fixed order (deterministic).
#pragma once
#include "qpx_grid.h"

namespace qpx {

struct TilePos {
    int tid, lane, w, g, c, A, B;   // B = -1: no second tile row
    QPX_DEV TilePos(const Block& blk, int nbl)
        : tid(blk.tid), lane(blk.lane()), w(blk.uniform(blk.wave())), g(blk.lane() >> 4), c(blk.lane() & 15),
          A(tile_row_a(nbl, blk.uniform(blk.wave()))), B(tile_row_b(nbl, blk.uniform(blk.wave())))
    {
    }
};

template <int NBL> struct TileMat {
    using T = double;
    static constexpr int NW = (NBL + 1) / 2, NT = 64 * NW, NSLOT = NBL | 1, MP = 16 * NBL;
    struct Pos : TilePos {
        QPX_DEV explicit Pos(const Block& blk) : TilePos(blk, NBL) {}
    };
    struct Regs { T e[NSLOT][4]; };
    // scratch: X (2 x 4 x MP) | S (2 x 16) | part (NW x NBL x 64) | yrow (MP)
    static constexpr int kX = 0, kS = 8 * MP, kPart = 8 * MP + 32, kRow = kPart + NW * NBL * 64;
    QPX_LAYOUT_HD static size_t scratch_elems() { return (size_t)kRow + MP; }
    static QPX_DEV void sync(const Block& blk)
    {
        if (NW == 1) blk.wave_sync();
        else blk.sync();
    }
    static QPX_DEV const T* image(const T* F, const FacLayout& lay) { return F + lay.Rm; }

    static QPX_DEV void load(const Block& blk, const Pos& p, Regs& E, const T* img)
    {
        const GlobalRows<T> rows(img, NW * NSLOT * 256, p.lane);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) E.e[s][r] = rows.row((p.w * NSLOT + s) * 4 + r);
    }

    static QPX_DEV void add_diag(const Pos& p, Regs& E, const T* vd)
    {
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            if (J == p.A) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p.c == p.g + 4 * r) E.e[J][r] += vd[16 * J + p.c];
            }
            if (J == p.B) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p.c == p.g + 4 * r) E.e[NSLOT - 1 - J][r] += vd[16 * J + p.c];
            }
        }
    }

    // sum over the 16 lanes of a DPP row; every lane ends with the total
    static QPX_DEV T rowsum16(const Block& blk, T v)
    {
        v += blk.template xor16<1>(v);
        v += blk.template xor16<2>(v);
        v += blk.template xor16<7>(v);
        v += blk.template xor16<15>(v);
        return v;
    }

    // out[j] = base[j] (if any) + the column partials of every wave that owns a tile in column j/16
    template <bool kNeg>
    static QPX_DEV void gather_cols(const Block& blk, const T* part, const T* base, T* out)
    {
        for (int j = blk.tid; j < MP; j += NT) {
            const int J = j >> 4, cc = j & 15;
            T sum = base[j];
            for (int w = 0; w < NW; ++w) {
                if (tile_row_a(NBL, w) < J) continue;
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
        T accA[4] = {0, 0, 0, 0}, accB[4] = {0, 0, 0, 0}, uA[4], uB[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uA[r] = vin[16 * p.A + p.g + 4 * r];
            uB[r] = p.B >= 0 ? vin[16 * p.B + p.g + 4 * r] : T(0);
        }
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            if (J > p.A) continue;
            const T xj = vin[16 * J + p.c];
            T col = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) accA[r] = fma_(E.e[J][r], xj, accA[r]);
            if (J < p.A) {
#pragma unroll
                for (int r = 0; r < 4; ++r) col = fma_(E.e[J][r], uA[r], col);
            }
            if (J <= p.B) {
#pragma unroll
                for (int r = 0; r < 4; ++r) accB[r] = fma_(E.e[NSLOT - 1 - J][r], xj, accB[r]);
                if (J < p.B) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) col = fma_(E.e[NSLOT - 1 - J][r], uB[r], col);
                }
            }
            part[(size_t)(p.w * NBL + J) * 64 + p.lane] = col;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            accA[r] = rowsum16(blk, accA[r]);
            if (NBL > 1) accB[r] = rowsum16(blk, accB[r]);
        }
        if (p.c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yrow[16 * p.A + p.g + 4 * r] = accA[r];
                if (p.B >= 0) yrow[16 * p.B + p.g + 4 * r] = accB[r];
            }
        }
        sync(blk);
        gather_cols<false>(blk, part, yrow, vout);
        sync(blk);
    }

    // ---- one panel of ldl_inv: rows/columns k0 .. k0+3, k0 = 16 Ip + 4 SP.  The tile row Ip is a
    // run-time (wave-uniform) value so that the code exists four times, not 4 NBL times; the register
    // index SP inside a tile is static.  gm[k] = (g == k) as 0/1.
    template <int SP>
    static QPX_DEV bool panel(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int Ip, const T (&gm)[4])
    {
        const int k0 = 16 * Ip + 4 * SP;
        T* X = scr + kX + (SP & 1) * 4 * MP;
        T* S = scr + kS + (SP & 1) * 16;
        const bool inpan = (p.c >> 2) == SP;
        const int kc = p.c & 3;
        const bool ownA = p.A == Ip, ownB = p.B == Ip;
        // -- publish: the panel's four rows left of the panel ...
        if (ownA || ownB) {
#pragma unroll
            for (int J = 0; J < NBL; ++J)
                if (J <= Ip) X[p.g * MP + 16 * J + p.c] = ownA ? E.e[J][SP] : E.e[NSLOT - 1 - J][SP];
        }
        // ... the pivot block (identity in X, the block itself in S) and the four columns below it.
        // Row g + 4 r of tile row I lies below the panel iff I > Ip or r > SP.  The panel's own columns
        // restart from zero: they are published, and the update writes -l~ W into them.
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            if (J != Ip) continue;
            if (inpan) {
                if (ownA || ownB) {
                    S[p.g * 4 + kc] = ownA ? E.e[J][SP] : E.e[NSLOT - 1 - J][SP];
                    X[p.g * MP + 16 * J + p.c] = (kc == p.g) ? T(1) : T(0);
                }
                if (p.A >= Ip) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (p.A > Ip || r > SP) X[kc * MP + 16 * p.A + p.g + 4 * r] = E.e[J][r];
                        E.e[J][r] = T(0);
                    }
                }
                if (p.B >= Ip) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (p.B > Ip || r > SP) X[kc * MP + 16 * p.B + p.g + 4 * r] = E.e[NSLOT - 1 - J][r];
                        E.e[NSLOT - 1 - J][r] = T(0);
                    }
                }
            }
        }
        sync(blk);
        // -- the 4 x 4 pivot block: S = L D L^T, W = L^-1 (unit lower), every lane the same numbers
        const T s00 = S[0], s10 = S[4], s11 = S[5], s20 = S[8], s21 = S[9], s22 = S[10];
        const T s30 = S[12], s31 = S[13], s32 = S[14], s33 = S[15];
        // operand reads that do not depend on the factorisation are issued before it
        const T* Xg = X + p.g * MP;
        T x0[NBL], x1[NBL], x2[NBL], xg[NBL];
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            const int j = 16 * J + p.c;
            x0[J] = X[j];
            x1[J] = X[MP + j];
            x2[J] = X[2 * MP + j];
            xg[J] = Xg[j];
        }
        const T d0 = s00, r0 = rcp_(d0);
        const T l10 = s10 * r0, l20 = s20 * r0, l30 = s30 * r0;
        const T d1 = fma_(-l10, s10, s11), r1 = rcp_(d1);
        const T v21 = fma_(-l20, s10, s21), v31 = fma_(-l30, s10, s31);
        const T l21 = v21 * r1, l31 = v31 * r1;
        const T d2 = fma_(-l21, v21, fma_(-l20, s20, s22)), r2 = rcp_(d2);
        const T v32 = fma_(-l31, v21, fma_(-l30, s20, s32));
        const T l32 = v32 * r2;
        const T d3 = fma_(-l32, v32, fma_(-l31, v31, fma_(-l30, s30, s33))), r3 = rcp_(d3);
        const bool good = (d0 > T(0)) && (d1 > T(0)) && (d2 > T(0)) && (d3 > T(0)) && finite_(d0) && finite_(d1) &&
                          finite_(d2) && finite_(d3);
        if (!good) return false;
        const T w10 = -l10, w21 = -l21, w32 = -l32;
        const T w20 = fma_(l21, l10, -l20);
        const T w31 = fma_(l32, l21, -l31);
        const T w30 = fma_(-w32, l20, fma_(-w31, l10, -l30));
        // row g of W, 1/d_g and d_g of this lane's group: sums against the 0/1 group masks (branch-free,
        // and cheaper than chains of 64-bit selects)
        const T cg0 = fma_(gm[3], w30, fma_(gm[2], w20, gm[1] * w10));
        const T cg1 = fma_(gm[3], w31, gm[2] * w21);
        const T cg2 = gm[3] * w32;
        const T rg = fma_(gm[3], r3, fma_(gm[2], r2, fma_(gm[1], r1, gm[0] * r0)));
        const T dg = fma_(gm[3], d3, fma_(gm[2], d2, fma_(gm[1], d1, gm[0] * d0)));
        if (p.w == 0 && p.c == 0) rd[k0 + p.g] = rg;
        // -- operands
        T bop[NBL];
        T tA = 0, tB = 0;
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            bop[J] = fma_(cg2, x2[J], fma_(cg1, x1[J], fma_(cg0, x0[J], xg[J])));
            if (J == p.A) tA = bop[J];
            if (J == p.B) tB = bop[J];
        }
        const bool right = p.c > 4 * SP + 3;          // rows of the panel's tile row below the panel
        const T aA = (p.A > Ip || (ownA && right)) ? -(tA * rg) : T(0);
        const T aB = (p.B > Ip || (ownB && right)) ? -(tB * rg) : T(0);
        // -- the panel's own rows are final
        if (ownA) {
#pragma unroll
            for (int J = 0; J < NBL; ++J)
                if (J <= Ip) E.e[J][SP] = (J == Ip && inpan && kc == p.g) ? dg : bop[J];
        }
        if (ownB) {
#pragma unroll
            for (int J = 0; J < NBL; ++J)
                if (J <= Ip) E.e[NSLOT - 1 - J][SP] = (J == Ip && inpan && kc == p.g) ? dg : bop[J];
        }
        // -- rank-4 update of every owned tile at or below the panel's tile row
        if (p.A >= Ip) {
#pragma unroll
            for (int J = 0; J < NBL; ++J)
                if (J <= p.A) blk.mfma16x16x4(aA, bop[J], E.e[J]);
        }
        if (p.B >= Ip) {
#pragma unroll
            for (int J = 0; J < NBL; ++J)
                if (J <= p.B) blk.mfma16x16x4(aB, bop[J], E.e[NSLOT - 1 - J]);
        }
        return true;
    }

    // E: T (SPD, order m, padded with the identity) -> strictly lower: W~ = L~^-1, diagonal: d_k;
    // rd[k] = 1/d_k.  Uniform return value (false: a pivot was not positive / finite).
    static QPX_DEV bool ldl_inv(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int m)
    {
        const T gm[4] = {p.g == 0 ? T(1) : T(0), p.g == 1 ? T(1) : T(0), p.g == 2 ? T(1) : T(0), p.g == 3 ? T(1) : T(0)};
        bool ok = true;
#pragma unroll 1
        for (int Ip = 0; Ip < NBL && ok; ++Ip) {
            const int k0 = 16 * Ip;
            if (k0 >= m) break;
            ok = panel<0>(blk, p, E, scr, rd, Ip, gm);
            if (ok && k0 + 4 < m) ok = panel<1>(blk, p, E, scr, rd, Ip, gm);
            if (ok && k0 + 8 < m) ok = panel<2>(blk, p, E, scr, rd, Ip, gm);
            if (ok && k0 + 12 < m) ok = panel<3>(blk, p, E, scr, rd, Ip, gm);
        }
        sync(blk);
        return ok;
    }

    // vout = -T^-1 vin = -W~^T D^-1 W~ vin (W~ unit lower in E, strictly lower part stored)
    static QPX_DEV void solve_neg(const Block& blk, const Pos& p, const Regs& E, const T* rd, int m, const T* vin,
                                  T* vout, T* tmp, T* scr)
    {
        T* part = scr + kPart;
        // u = D^-1 W~ vin: sums along tile rows, complete inside the owning wave
        T accA[4] = {0, 0, 0, 0}, accB[4] = {0, 0, 0, 0};
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            if (J > p.A) continue;
            const T xj = vin[16 * J + p.c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T e = (J < p.A || p.c < p.g + 4 * r) ? E.e[J][r] : T(0);
                accA[r] = fma_(e, xj, accA[r]);
            }
            if (J <= p.B) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T e = (J < p.B || p.c < p.g + 4 * r) ? E.e[NSLOT - 1 - J][r] : T(0);
                    accB[r] = fma_(e, xj, accB[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            accA[r] = rowsum16(blk, accA[r]);
            if (NBL > 1) accB[r] = rowsum16(blk, accB[r]);
        }
        if (p.c == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ia = 16 * p.A + p.g + 4 * r;
                tmp[ia] = (ia < m) ? (accA[r] + vin[ia]) * rd[ia] : T(0);
                if (p.B >= 0) {
                    const int ib = 16 * p.B + p.g + 4 * r;
                    tmp[ib] = (ib < m) ? (accB[r] + vin[ib]) * rd[ib] : T(0);
                }
            }
        }
        sync(blk);
        // x = W~^T u: sums down columns, partial per wave, gathered in a fixed order
        T uA[4], uB[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uA[r] = tmp[16 * p.A + p.g + 4 * r];
            uB[r] = p.B >= 0 ? tmp[16 * p.B + p.g + 4 * r] : T(0);
        }
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            if (J > p.A) continue;
            T col = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T e = (J < p.A || p.c < p.g + 4 * r) ? E.e[J][r] : T(0);
                col = fma_(e, uA[r], col);
            }
            if (J <= p.B) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T e = (J < p.B || p.c < p.g + 4 * r) ? E.e[NSLOT - 1 - J][r] : T(0);
                    col = fma_(e, uB[r], col);
                }
            }
            part[(size_t)(p.w * NBL + J) * 64 + p.lane] = col;
        }
        sync(blk);
        gather_cols<true>(blk, part, tmp, vout);
        sync(blk);
    }
};

QPX_LAYOUT_HD size_t lds_elems_ipm_tile(int nbl, int n, int q)
{
    const size_t mp = 16 * (size_t)nbl, nw = (size_t)tile_nw(nbl);
    return lds_elems_ipm_loop(mp, 8 * mp + 32 + nw * nbl * 64 + mp, n, q);
}

template <int NBL, int NS>
QPX_DEV void ipm_tile_body(const Block& b, const IpmArgs<double>& a, int qp, double* lds)
{
    ipm_loop_body<double, TileMat<NBL>, NS>(b, a, qp, lds);
}

}  // namespace qpx
