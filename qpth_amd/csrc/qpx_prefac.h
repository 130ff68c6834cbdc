// qpx_prefac.h -- pre_factor_kkt on the MATRIX CORES (f64, 33 <= nz + neq <= 112, nineq <= 112; round 4).
// (Described for neq = 0; equality constraints: the note at prefac_tile_body.)
//
// Replaces, at those sizes, the symmetric sweep of the augmented matrix (qpx_grid.h: sweep_body; reference:
// pre_factor_kkt, qpth/solvers/pdipm/batch.py:375-429) and writes the SAME blob (qpx_layout.h, family (a)): -K = -Q^-1
// (both triangles), M^T = K G^T, R = G K G^T as the tile image, || G^T 1 ||.  The sweep is bound by LDS bandwidth and
// vector FMAs (0.157 ms at C2); here everything but one factorisation is tile products on v_mfma_f64_16x16x4:
//
//   P0  Q = L~ D L~^T with W~ = L~^-1 in place (TileMat<NBN, 4, true>::ldl_inv: the loop kernel's chain-wave
//       factorisation, one launch-mate wave eliminating the pivot blocks a panel ahead of the three that hold the tiles),
//       then V = D^-1/2 W~ (lower triangular, Q^-1 = V^T V) -> LDS, tile by tile, rows 17 apart.
//   P1  every wave owns one or two blocks of sixteen rows of G ("m-blocks", dealt from the last): for block i
//         Yt[k, i] = sum_{k' <= k} V[k, k'] G^T[k', i]     (Yt = V G^T, n x m; A operand from LDS, B = G from global,
//                                                          read once into registers in the operand layout)
//         M^T[j, i] = sum_{k >= j} V[k, j]^T Yt[k, i]      (A from LDS, B = the accumulator tiles of Yt as they stand:
//                                                          register r of a tile in the C/D layout IS the B operand of
//                                                          the k-slice of rows g + 4 r, qpx_tile.h) -> global
//       and a share of  K[j1, j2] = sum_{k >= j1} V[k, j1]^T V[k, j2]  (both operands from LDS), dealt over the waves
//       so that their tile-product counts come out equal.
//   P2  V is dead: the blocks of Yt move to LDS in the accumulator layout (what fits: five blocks of seven tiles; the
//       last two stay in their owners' registers) and  R[i1, i2] = sum_k Yt[k, i1]^T Yt[k, i2]  is dealt over the waves
//       the same way (A: registers of the owner, or LDS; B: LDS), straight into the tile image.
// Every product contracts over the n index, so k-slices of four that are identity padding (n = 100: three of 28) are
// skipped.  At C2: 2 352 matrix instructions per QP instead of the sweep's ~190 k vector FMAs per thread-row.
// Q is read by its upper triangle and G as it is, like the sweep; failure of a pivot = `Q is not SPD`, blob zeroed.
#pragma once
#include <type_traits>

#include "qpx_tile.h"

namespace qpx {

constexpr int kPfVS = 16 * 17;                          // one tile of V in LDS
QPX_LAYOUT_HD constexpr int prefac_cap(int nbn) { return nbn == 7 ? 5 : 7; }     // blocks of Yt that fit in LDS beside the rest
QPX_LAYOUT_HD constexpr size_t prefac_fixed_elems(int nbn) { return (size_t)16 * nbn + 8 + 7 * 16 * (size_t)nbn; }
// nbm: the m-blocks there are (blocks of Yt beyond them are never staged).  (r6: sized by prefac_cap alone the kernel took
// 61 KB at four tile rows whatever nineq -- one workgroup per 80 KB half of a CU's LDS; with nineq <= 64 it is 40.9 KB, two.)
QPX_LAYOUT_HD constexpr size_t lds_elems_prefac_tile(int nbn, int nbm = 1 << 20)
{
    const int slots = nbm < prefac_cap(nbn) ? nbm : prefac_cap(nbn);
    const size_t scr = tile_scratch_elems(nbn, 3, true), vs = (size_t)(nbn * (nbn + 1) / 2) * kPfVS,
                 yb = (size_t)slots * nbn * 256;
    const size_t u = scr > vs ? (scr > yb ? scr : yb) : (vs > yb ? vs : yb);
    return prefac_fixed_elems(nbn) + u;
}
// does the matrix-core pre-factorisation serve this size?  (images: fac_layout's, 4 = the tile image only)
// (neq > 0, round 4: the factorisation is of the quasi-definite [[Q, A^T], [A, 0]] of order nz + neq <= 112)
QPX_LAYOUT_HD bool prefac_tile_serves(int n, int m, int q, int images)
{
    return images == 4 && n + q <= 112 && (tile_nb(n + q) == 4 || tile_nb(n + q) == 7) && tile_nb(m) > 0;
}

// Which wave computes which tile of K and of R (see prefac_tile_body; host side, once per launch): the wave with the
// least tile products so far takes the next tile (ties: the highest wave, which owns the fewest m-blocks).  In the kernel
// this greedy loop cost every wave ~1 000 scalar instructions, i.e. ~5 k cycles on its critical path.
inline void prefac_deal(int nbn, int n, int m, unsigned (&pf_k)[4], unsigned (&pf_r)[4])
{
    const int nbm = (m + 15) >> 4, nbr = (n + 15) >> 4, cap = prefac_cap(nbn);
    const int kextra = 1;          // a tile of K costs a little more than its products (its own loop, the mirrored stores); 0 .. 9 measured equal within noise (profiles/archive/r04q)
    int L[4];
    auto pick = [&](int cost) {
        int ww = 3;
        for (int x = 2; x >= 0; --x)
            if (L[x] < L[ww]) ww = x;
        L[ww] += cost;
        return ww;
    };
    for (int w = 0; w < 4; ++w) {
        pf_k[w] = pf_r[w] = 0;
        L[w] = ((nbm - 1 - w >= 0 ? 1 : 0) + (nbm - 5 - w >= 0 ? 1 : 0)) * nbn * (nbn + 1);      // tile products of its m-blocks (Yt and M^T)
    }
    for (int j1 = 0; j1 < nbr; ++j1)
        for (int j2 = 0; j2 <= j1; ++j2) pf_k[pick(nbn - j1 + kextra)] |= 1u << (j1 * (j1 + 1) / 2 + j2);
    // R: the rows of the blocks that stay in registers belong to their owners (waves 0 and 1)
    const int nst = nbm < cap ? nbm : cap;
    L[0] = L[1] = L[2] = L[3] = 0;
    if (nbm > cap) {
        L[0] = (nbm - cap == 2) ? nst + 2 : nst + 1;
        if (nbm - cap == 2) L[1] = nst + 1;
    }
    for (int i1 = 0; i1 < nst; ++i1)
        for (int i2 = 0; i2 <= i1; ++i2) pf_r[pick(1)] |= 1u << (i1 * (i1 + 1) / 2 + i2);
}

// kEq (neq > 0): the same kernel on the augmented matrix Ka = [[Q, A^T], [A, 0]] of order nn = n + q <= 112.  Its
// un-pivoted factorisation Ka = L~ D L~^T has q negative pivots (-A Q^-1 A^T, the sweep's second batch of pivots); with
// V = |D|^-1/2 L~^-1 and S = sign(D),  Ka^-1 = V^T S V = [[K, N], [N^T, -S11^-1]]  and with Yt = V [G^T; 0]
//   [M^T; W^T] = V^T S Yt,   R = Yt^T S Yt
// -- the products of the neq = 0 kernel with the rows n .. nn - 1 of one operand negated, and three more arrays of the
// blob read off the rows beyond n of the results: -N^T and S11^-1 from the tiles of Ka^-1, W^T from the rows of M^T.
template <int NBN, bool kEq = false>
QPX_DEV void prefac_tile_body(const Block& b, const PrefactorArgs<double>& a, int qp, double* lds)
{
    using T = double;
    using TM = TileMat<NBN, 4, true>;
    constexpr int MPN = 16 * NBN, CAP = prefac_cap(NBN), VS = kPfVS, YBS = NBN * 256;
    const int n = a.n, m = a.m, q = kEq ? a.q : 0, nn = n + q;          // nn: order of the factorisation
    const FacLayout lay = fac_layout(n, m, q, a.images);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const In<T> Qg(a.Q, (size_t)qp * a.sQ, a.io32), Gg(a.G, (size_t)qp * a.sG, a.io32);
    T* rd = lds;                       // 1 / d_k of the factorisation (MPN)
    T* flag = rd + MPN;                // 8
    T* gpart = flag + 8;               // column sums of G per m-block (7 x MPN)
    T* un = gpart + 7 * MPN;           // the factorisation's scratch | V | the staged blocks of Yt
    T* Vs = un;
    T* Yb = un;
    const int nbm = (m + 15) >> 4;     // m-blocks
    const int nbr = (nn + 15) >> 4;    // tile rows of V that are not all padding
    auto sgn = [n, nn](int row) { return (kEq && row >= n && row < nn) ? T(-1) : T(1); };      // S
    auto nsl = [nn](int k) { const int left = nn - 16 * k; return left >= 16 ? 4 : (left <= 0 ? 0 : (left + 3) >> 2); };

    QPX_PROF_INIT
    // ---- P0: V = D^-1/2 L~^-1 of Q
    // Q's upper triangle -> LDS (row j from column j on, packed), by all threads and coalesced; the tile waves pick
    // their tiles out of it.  (Read straight into the tile registers -- sixteen rows x 32 bytes per load -- the forty
    // loads of a wave were forty serialised round trips to memory when guarded by lane conditions, and cost the
    // factorisation behind them 500 spilled registers when not.)
    {
        T* qs = un;
        T* dump = gpart + b.tid;                      // where a lane left of the diagonal / beyond n stores (one slot per thread; 256 <= 7 MPN)
        auto stage_q = [&](auto tag) {
            using S = decltype(tag);
            const GlobalBuf<S> qb(reinterpret_cast<const S*>(a.Q) + (size_t)qp * a.sQ, (long long)n * n);
            const int col = b.tid & 127, r0 = b.tid >> 7;
            // every load and every store unconditional (a load whose only use is a guarded store is sunk into the guard
            // by the compiler and waited for there: one round trip to memory per row); a lane that has nothing to fetch
            // re-reads its row's diagonal element, which costs no traffic
            T v[8 * NBN];
#pragma unroll
            for (int u = 0; u < 8 * NBN; ++u) {
                const int row = r0 + 2 * u;
                const bool ok = row < n && col >= row && col < n;
                v[u] = qb.at(ok ? row * n + col : (row < n ? row * n + row : 0));
            }
#pragma unroll
            for (int u = 0; u < 8 * NBN; ++u) {
                const int row = r0 + 2 * u;
                const bool ok = row < n && col >= row && col < n;
                T* dst = ok ? qs + (row * nn - (row * (row - 1)) / 2 + col - row) : dump;
                *dst = v[u];
            }
        };
        if (a.io32) stage_q(float());
        else stage_q(double());
        if constexpr (kEq) {
            // A^T right of Q (row r of the packed triangle, columns n .. nn - 1), zeros below it
            auto stage_a = [&](auto tag) {
                using S = decltype(tag);
                const GlobalBuf<S> ab(reinterpret_cast<const S*>(a.A) + (size_t)qp * a.sA, (long long)q * n);
                constexpr int NA = (MPN * MPN / 4 + 255) / 256;       // q n <= (nn / 2)^2
                T v[NA];
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    const int e = b.tid + 256 * u;
                    v[u] = ab.at(e < q * n ? e : 0);
                }
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    const int e = b.tid + 256 * u, j = e / n, r = e - j * n;          // A[j][r]
                    T* dst = e < q * n ? qs + (r * nn - (r * (r - 1)) / 2 + n + j - r) : dump;
                    *dst = v[u];
                }
            };
            if (a.io32) stage_a(float());
            else stage_a(double());
            for (int e = b.tid; e < q * (q + 1) / 2; e += b.nt) qs[n * nn - (n * (n - 1)) / 2 + e] = T(0);    // rows n .. nn - 1
        }
        b.sync();
    }
    // the tile image of R: zero where nothing is written below (rows / columns of padding, tile rows beyond nbm); behind
    // the loads of Q in the memory pipeline, drained under the factorisation, barriers away from the stores of R
    for (size_t e = b.tid; e < tile_image_elems(lay.nbt); e += b.nt) F[lay.Rm + e] = T(0);
    QPX_PROF(0)
    const typename TM::Pos p0(b);
    TM::with_role(p0, [&](const auto& gp) {
        using P = typename std::decay<decltype(gp)>::type;
        const P p = gp.fresh();
        typename TM::Regs E;
        {
            const T* qs = un;
#pragma unroll
            for (int pp = 0; pp < TM::NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int J = 0; J < TM::psize(pp); ++J) {
                    if (J <= I) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * I + p.g + 4 * r, j = 16 * J + p.c;
                            const int lo = i < j ? i : j, hi = i < j ? j : i;       // Q by its upper triangle, like the sweep
                            const T v = qs[hi < nn ? lo * nn - (lo * (lo - 1)) / 2 + hi - lo : 0];
                            E.e[TM::slot(pp, J)][r] = hi < nn ? v : (i == j ? T(1) : T(0));
                        }
                    }
                }
            }
        }
        b.sync();                                    // the staged Q lies where the factorisation's scratch does
        QPX_PROF(1)
        int fcode = 0;                               // 0, or QPX_ST_Q_NOT_SPD / QPX_ST_A_RANK (= the flag of the pivot block)
        if constexpr (kEq) {
            b.template prio<3>();
            fcode = TM::template factor_role<TM::template role_of<P>::value, false, kEq>(b, p, E, un, rd, nbr < NBN ? nbr : NBN, nn, [n, nn](int k) {
                const int kmax = nn - 16 * k, pos = n - 16 * k;             // pivots of the block; how many of them are Q's
                return typename TM::PanelOf{kmax, pos >= 16 || pos >= kmax ? 1 : (pos <= 0 ? -1 : 2 + pos)};
            });
        } else {
            fcode = TM::ldl_inv(b, p, E, un, rd, n) ? 0 : 1;
        }
        const bool ok = fcode == 0;
        b.sync();                                    // the factorisation's scratch is dead: V goes on top of it
        QPX_PROF(2)
        // D^-1/2 once per row (the square root is a twenty-instruction sequence: per element of V it cost more than the
        // factorisation's last panel)
        T* rsq = gpart;
        for (int i = p.tid; i < MPN; i += TM::NT) rsq[i] = (ok && i < nn) ? sqrt_(kEq ? abs_(rd[i]) : rd[i]) : T(1);
        b.sync();
        if (p.is_chain()) {
            if (p.lane == 0) flag[0] = (T)fcode;
        } else if (ok) {
#pragma unroll
            for (int pp = 0; pp < TM::NPOS; ++pp) {
                const int I = p.row(pp);
                T rs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rs[r] = I >= 0 ? rsq[16 * I + p.g + 4 * r] : T(1);
#pragma unroll
                for (int J = 0; J < TM::psize(pp); ++J) {
                    if (J <= I) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * I + p.g + 4 * r, j = 16 * J + p.c;
                            const T e = E.e[TM::slot(pp, J)][r];
                            const T lowv = i < nn ? e * rs[r] : T(0);            // (rows of padding: the identity)
                            Vs[(I * (I + 1) / 2 + J) * VS + (p.g + 4 * r) * 17 + p.c] = j < i ? lowv : (j == i ? rs[r] : T(0));
                        }
                    }
                }
            }
        }
        b.sync();
        QPX_PROF(3)
    });
    if (flag[0] != T(0)) {
        for (size_t e = b.tid; e < lay.total; e += b.nt) F[e] = T(0);
        if (b.tid == 0) a.status[qp] = flag[0] == T(2) ? QPX_ST_A_RANK : QPX_ST_Q_NOT_SPD;
        return;
    }

    // ---- P1
    // (Scheduling: the products below are straight-line code -- operands of a group fetched from LDS ahead of its matrix
    // instructions, no branch per instruction.  The first version guarded every instruction by "is this k-slice
    // padding" and the compiler answered with branch / ds_read / wait / v_mfma quadruples: ~200 cycles per instruction.
    // Now only the slices of the LAST tile row are guarded (all other padding computes zeros).)
    b.template prio<0>();
    const int w = b.uniform(b.wave());
    int lane = b.lane(), g = lane >> 4, c = lane & 15;
    QPX_LAUNDER_V(lane);
    QPX_LAUNDER_V(g);
    QPX_LAUNDER_V(c);
    const int iA = nbm - 1 - w, iB = nbm - 5 - w;            // this wave's m-blocks (< 0: none)
    const int nsl_ = nsl(NBN - 1);                           // k-slices of the last tile row that are not padding (0 .. 4)
    T YA[NBN][4], YB[NBN][4];

    // G rows 16 i .. 16 i + 15 in the operand layout: lane (g, c) holds G(16 i + c, 16 k' + 4 g + s) for slice s -- which
    // four columns of a k-slice go to which lane group is free as long as the A operand follows, and this way a lane's
    // four values are 32 consecutive bytes (two 16-byte loads).  (With column 4 s + g every 128-byte line was touched by
    // four load instructions of the wave, and with eight waves' blocks thrashing a 16 KB L1 each touch came from the L2:
    // the workgroup's memory pipeline was busy with G for ~10 k cycles, profiles/archive/r04p.)
    // (buffer loads from the block's first row: one lane offset for the 28 loads, rows beyond m read as zero; columns
    // beyond n are masked where the values are used, so that nothing waits for the loads here)
    auto load_g = [&](int i, T (&Gop)[NBN][4]) {
        auto body = [&](auto tag) {
            using S = decltype(tag);
            const GlobalBuf<S> gb(reinterpret_cast<const S*>(a.G) + (size_t)qp * a.sG + (size_t)16 * i * n, (long long)(m - 16 * i) * n);
            const int voff = c * n + 4 * g;
#pragma unroll
            for (int kp = 0; kp < NBN; ++kp) {
                T lo[2], hi[2];
                gb.at2(voff + 16 * kp, lo);
                gb.at2(voff + 16 * kp + 2, hi);
                Gop[kp][0] = lo[0]; Gop[kp][1] = lo[1]; Gop[kp][2] = hi[0]; Gop[kp][3] = hi[1];
            }
        };
        if (a.io32) body(float());
        else body(double());
    };
    auto mask_g = [&](T (&Gop)[NBN][4]) {
#pragma unroll
        for (int kp = 0; kp < NBN; ++kp)
#pragma unroll
            for (int s = 0; s < 4; ++s) Gop[kp][s] = (16 * kp + 4 * g + s < n) ? Gop[kp][s] : T(0);
    };
    // column sums of the block (for || G^T 1 ||) out of the operand registers: sums over the sixteen lanes of a row
    auto colsum_g = [&](int i, const T (&Gop)[NBN][4]) {
#pragma unroll
        for (int kp = 0; kp < NBN; ++kp)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                T x = Gop[kp][s];
                x += b.template xor16<1>(x);
                x += b.template xor16<2>(x);
                x += b.template xor16<7>(x);
                x += b.template xor16<15>(x);
                if (c == 0) gpart[i * MPN + 16 * kp + 4 * g + s] = x;
            }
    };
    auto y_block = [&](const T (&Gop)[NBN][4], T (&Y)[NBN][4]) {
#pragma unroll
        for (int k = 0; k < NBN; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[k][r] = T(0);
#pragma unroll
        for (int kp = 0; kp < NBN; ++kp) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (kp == NBN - 1 && nsl_ == 0) continue;               // (uniform; a slice here is columns 4 g + s: all four or none)
                T av[NBN];
#pragma unroll
                for (int k = kp; k < NBN; ++k) av[k] = Vs[(k * (k + 1) / 2 + kp) * VS + c * 17 + 4 * g + s];
#pragma unroll
                for (int k = kp; k < NBN; ++k) b.mfma16x16x4(av[k], Gop[kp][s], Y[k]);
            }
        }
    };
    // kEq: from here on a block of Yt is kept as S Yt (rows n .. nn - 1 negated): M^T = V^T (S Yt) as it stands, and in
    // R = Yt^T S Yt = (S (S Yt))^T (S Yt) the A operand gets the sign back (lds_product, r_tile_reg)
    auto sign_y = [&](T (&Y)[NBN][4]) {
        if constexpr (kEq) {
            int gg = g;
            QPX_LAUNDER_V(gg);          // (the 28 row predicates are recomputed where they are used: kept, they are 56 scalar registers + spills)
#pragma unroll
            for (int k = 0; k < NBN; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) Y[k][r] *= sgn(16 * k + gg + 4 * r);
        }
    };
    auto mt_block = [&](int i, const T (&Y)[NBN][4]) {
#pragma unroll
        for (int j0 = 0; j0 < NBN; j0 += 2) {
            T acc0[4] = {T(0), T(0), T(0), T(0)}, acc1[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int k = j0; k < NBN; ++k) {
                T a0[4], a1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0[r] = Vs[(k * (k + 1) / 2 + j0) * VS + (g + 4 * r) * 17 + c];
                    if (j0 + 1 < NBN && k >= j0 + 1) a1[r] = Vs[(k * (k + 1) / 2 + j0 + 1) * VS + (g + 4 * r) * 17 + c];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (k == NBN - 1 && r >= nsl_) continue;            // (uniform; the last tile row only)
                    b.mfma16x16x4(a0[r], Y[k][r], acc0);
                    if (j0 + 1 < NBN && k >= j0 + 1) b.mfma16x16x4(a1[r], Y[k][r], acc1);
                }
            }
            const int col = 16 * i + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int r0 = 16 * j0 + g + 4 * r, r1 = r0 + 16;
                if (col < m && r0 < n) F[lay.MT + (size_t)r0 * m + col] = acc0[r];
                if (j0 + 1 < NBN && col < m && r1 < n) F[lay.MT + (size_t)r1 * m + col] = acc1[r];
                if constexpr (kEq) {                 // rows n .. nn - 1 of [M^T; W^T]: W = G N, m x q
                    if (col < m && r0 >= n && r0 < nn) F[lay.W + (size_t)col * q + (r0 - n)] = acc0[r];
                    if (j0 + 1 < NBN && col < m && r1 >= n && r1 < nn) F[lay.W + (size_t)col * q + (r1 - n)] = acc1[r];
                }
            }
        }
    };
    // one tile product with both operands in LDS: pa / pb = this lane's element of slice 0 of tile row k0, `step(k)` =
    // distance to tile row k; tile rows k0 .. NBN - 1.  The operands of row k + 1 are fetched before the matrix
    // instructions of row k are issued; two accumulators (chains of two dependent instructions instead of four).
    auto lds_product = [&](int k0, auto&& ofs_a, auto&& ofs_b, int rs, T (&acc)[4]) {
        T acc2[4] = {T(0), T(0), T(0), T(0)};
        T av[4], bv[4], an[4], bn[4];
        auto fetch = [&](int k, T (&x)[4], T (&y)[4]) {
            const T* va = ofs_a(k);
            const T* vb = ofs_b(k);
#pragma unroll
            for (int r = 0; r < 4; ++r) { x[r] = va[rs * r]; y[r] = vb[rs * r]; }
            if constexpr (kEq) {
                int gg = g;
                QPX_LAUNDER_V(gg);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] *= sgn(16 * k + gg + 4 * r);
            }
        };
        fetch(k0, av, bv);
#pragma unroll 1
        for (int k = k0; k < NBN - 1; ++k) {
            fetch(k + 1, an, bn);
            b.mfma16x16x4(av[0], bv[0], acc);
            b.mfma16x16x4(av[1], bv[1], acc2);
            b.mfma16x16x4(av[2], bv[2], acc);
            b.mfma16x16x4(av[3], bv[3], acc2);
#pragma unroll
            for (int r = 0; r < 4; ++r) { av[r] = an[r]; bv[r] = bn[r]; }
        }
        if (nsl_ > 0) b.mfma16x16x4(av[0], bv[0], acc);
        if (nsl_ > 1) b.mfma16x16x4(av[1], bv[1], acc2);
        if (nsl_ > 2) b.mfma16x16x4(av[2], bv[2], acc);
        if (nsl_ > 3) b.mfma16x16x4(av[3], bv[3], acc2);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
    };
    auto k_tile = [&](int j1, int j2) {
        T acc[4] = {T(0), T(0), T(0), T(0)};
        lds_product(j1, [&](int k) { return Vs + (k * (k + 1) / 2 + j1) * VS + g * 17 + c; },
                    [&](int k) { return Vs + (k * (k + 1) / 2 + j2) * VS + g * 17 + c; }, 68, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * j1 + g + 4 * r, j = 16 * j2 + c;
            if (i < n && j < n) {
                F[lay.Kneg + (size_t)i * n + j] = -acc[r];
                if (j1 != j2) F[lay.Kneg + (size_t)j * n + i] = -acc[r];
            }
            if constexpr (kEq) {                     // rows n .. nn - 1 of Ka^-1: [N^T, -S11^-1]
                if (i >= n && i < nn && j < n) F[lay.NTn + (size_t)(i - n) * n + j] = -acc[r];
                if (i >= n && i < nn && j >= n && j < nn) {
                    F[lay.S11i + (size_t)(i - n) * q + (j - n)] = -acc[r];
                    if (j1 != j2) F[lay.S11i + (size_t)(j - n) * q + (i - n)] = -acc[r];
                }
            }
        }
    };
    // tiles of a lower block triangle by number: t = i (i + 1) / 2 + j
    auto tri_row = [](int t) { return (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15) + (t >= 21); };
    auto my_mask = [&](const unsigned (&mk)[4]) { return w == 0 ? mk[0] : (w == 1 ? mk[1] : (w == 2 ? mk[2] : mk[3])); };
    T Gop[NBN][4];
    if (iA >= 0) load_g(iA, Gop);
#pragma unroll 1
    for (unsigned km = my_mask(a.pf_k); km != 0; km &= km - 1) {
        const int t = __builtin_ctz(km), j1 = tri_row(t);
        k_tile(j1, t - j1 * (j1 + 1) / 2);
    }
    QPX_PROF(4)
    if (iA >= 0) {
        mask_g(Gop);
        y_block(Gop, YA);
        sign_y(YA);
        colsum_g(iA, Gop);
        if (iB >= 0) load_g(iB, Gop);
        mt_block(iA, YA);
        if (iB >= 0) {
            mask_g(Gop);
            y_block(Gop, YB);
            sign_y(YB);
            colsum_g(iB, Gop);
            mt_block(iB, YB);
        }
    }
    QPX_PROF(5)
    b.sync();                                                 // V is dead
    // ---- P2
    auto stage = [&](const T (&Y)[NBN][4], int slot) {
#pragma unroll
        for (int k = 0; k < NBN; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) Yb[slot * YBS + k * 256 + r * 64 + lane] = Y[k][r];
    };
    if (iA >= 0 && iA < CAP) stage(YA, iA);
    if (iB >= 0 && iB < CAP) stage(YB, iB);
    if (w == 3) {
        T acc = T(0);
        for (int j = lane; j < n; j += kWave) {
            T s = T(0);
            for (int i = 0; i < nbm; ++i) s += gpart[i * MPN + j];
            acc = fma_(s, s, acc);
        }
        acc = wave_sum(b, acc);
        if (lane == 0) F[lay.scal] = sqrt_(acc);
    }
    b.sync();
    QPX_PROF(6)
    auto r_store = [&](int i1, int i2, const T (&acc)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) F[lay.Rm + (size_t)((i1 * (i1 + 1) / 2 + i2) * 4 + r) * 64 + lane] = acc[r];
    };
    auto r_tile_lds = [&](int i1, int i2) {
        T acc[4] = {T(0), T(0), T(0), T(0)};
        lds_product(0, [&](int k) { return Yb + i1 * YBS + k * 256 + lane; }, [&](int k) { return Yb + i2 * YBS + k * 256 + lane; }, 64, acc);
        r_store(i1, i2, acc);
    };
    // A operand from this wave's registers (block i1), B from LDS slot `slot` (or, slot < 0, the same registers)
    auto r_tile_reg = [&](const T (&Y)[NBN][4], int i1, int i2, int slot) {
        T acc[4] = {T(0), T(0), T(0), T(0)}, acc2[4] = {T(0), T(0), T(0), T(0)};
        const T* yb = Yb + (slot < 0 ? 0 : slot) * YBS + lane;
        int gg = g;
        QPX_LAUNDER_V(gg);
#pragma unroll
        for (int k = 0; k < NBN; ++k) {
            T bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = slot < 0 ? Y[k][r] : yb[k * 256 + r * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k == NBN - 1 && r >= nsl_) continue;                // (uniform; the last tile row only)
                const T av = kEq ? Y[k][r] * sgn(16 * k + gg + 4 * r) : Y[k][r];
                if (r & 1) b.mfma16x16x4(av, bv[r], acc2);
                else b.mfma16x16x4(av, bv[r], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
        r_store(i1, i2, acc);
    };
    const int nst = nbm < CAP ? nbm : CAP;                    // staged blocks
    if (nbm > CAP && iA >= CAP) {                             // rows of the blocks that stayed in registers: by their owners
#pragma unroll 1
        for (int i2 = 0; i2 < nst; ++i2) r_tile_reg(YA, iA, i2, i2);
        r_tile_reg(YA, iA, iA, -1);
    }
#pragma unroll 1
    for (unsigned rm = my_mask(a.pf_r); rm != 0; rm &= rm - 1) {
        const int t = __builtin_ctz(rm), i1 = tri_row(t);
        r_tile_lds(i1, t - i1 * (i1 + 1) / 2);
    }
    if (nbm - CAP == 2) {                                     // the tile of the two register-held blocks: one of them through LDS
        b.sync();
        if (w == 1) stage(YA, 0);
        b.sync();
        if (w == 0) r_tile_reg(YA, nbm - 1, nbm - 2, 0);
    }
    QPX_PROF(7)
    QPX_PROF_DUMP(F + lay.prof, T)
    if (b.tid == 0) a.status[qp] = 0;
}

}  // namespace qpx
