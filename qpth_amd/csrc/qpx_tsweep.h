// qpx_tsweep.h -- the pre-factorisation as a symmetric sweep ON MATRIX-CORE TILES (f64, gfx950; replaces
// pre_factor_kkt, batch.py:375-429, and the rank-1 thread-grid sweep of qpx_grid.h at the sizes it is built for).
//
// The augmented matrix  S = [[Q, A^T, G^T], [A, 0, 0], [G, 0, 0]]  is laid out in tiles of 16 with EVERY block padded to
// a multiple of 16 (identity on the padded diagonal of the Q block: decoupled unit pivots that are skipped), so that a
// sixteen-column panel never straddles two blocks:  NQ tile rows of Q | NA of A | NG of G  (+ identity rows up to the
// instantiated NBL).  Sweeping the NQ + NA pivot panels -- positive pivots in Q, negative ones in the equality
// block -- leaves
//      [[ -K,   .,       . ],
//       [ -N^T, S_A^-1,  . ],          K = Q^-1 - Q^-1 A^T S_A^-1 A Q^-1,  N = Q^-1 A^T S_A^-1,  S_A = A Q^-1 A^T,
//       [ -M,   -W,     -R ]]          M = G K,  W = G N,  R = G K G^T  (the reference's R)
// i.e. the blob of fac_layout (qpx_layout.h) up to signs and one transposition.
//
// The sweep of a sixteen-pivot panel P is the blocked elimination of qpx_tile.h with two changes (TileMat::update_row,
// kSweep): with X_J the panel's sixteen old rows, S_PP = L~ D L~^T the pivot block, b_J = L~^-1 X_J (b_P = L~^-1),
//      every tile (I, J), I != P:  E(I, J) += (-D^-1 b_I)^T b_J      (rows above the panel too: they accumulate -S11^-1)
//      the panel's own row:        E(P, J)  = (-D^-1 b_P)^T b_J      (J = P: -S_PP^-1)
// so every tile of the lower block triangle gets one rank-16 update (four MFMAs) per panel -- uniform work.  NBL is
// even and tile rows are dealt in pairs: tile wave w (of NBL / 2) owns rows NBL-1-w and w, i.e. NBL + 1 tiles whatever w
// is -- "slot" s of every tile wave holds a tile, and which one is a wave-uniform scalar.  ALL TILE WAVES THEREFORE RUN
// THE SAME STRAIGHT-LINE CODE (LDS addresses differ): no per-role copies of the kernel.  That matters here more than in
// the loop kernel, whose body runs a dozen times: every instruction of this kernel runs once or seven times, and its
// first version (one body per wave role, 200 KB of text) spent more time fetching instructions than computing.
// One more wave, the chain wave, eliminates the pivot block of panel P+1 while the tile waves stream panel P
// (qpx_tile.h, chain-wave form).  One workgroup per QP, one QP per CU at a time: at C2 (NBL = 14, 105 tiles, 7 panels)
// 3300 MFMAs per QP against 100 barrier-separated rank-1 steps in the thread-grid sweep.
#pragma once
#include "qpx_tile.h"

namespace qpx {

// tile rows of the padded augmented matrix, and the instantiated size that serves them (0: none)
QPX_LAYOUT_HD int tsweep_need(int n, int m, int q) { return (n + 15) / 16 + (q + 15) / 16 + (m + 15) / 16; }
// LDS: 1/d (16 nbl) | the tile kernels' scratch (X, S, W, flag, BT + AT, row buffer, S2), which doubles as the staging
// area between global memory and the tiles | the column sums of G (4 x 16 nbl)
QPX_LAYOUT_HD constexpr size_t tsweep_scratch_elems(int nbl) { return tile_scratch_elems(nbl, 7, true); }
QPX_LAYOUT_HD size_t lds_elems_tsweep(int nbl) { return 16 * (size_t)nbl + tsweep_scratch_elems(nbl) + 64 * (size_t)nbl; }
QPX_LAYOUT_HD int tsweep_nb(int n, int m, int q)
{
    const int need = tsweep_need(n, m, q);
    const int nbl = need <= 8 ? 8 : (need <= 12 ? 12 : (need <= 14 ? 14 : 0));
    if (nbl == 0) return 0;
    // the staging rounds must fit the scratch region: NP (NP + 1) / 2 tiles (pivot rows), NG x NP tiles (G rows)
    const int NP = (n + 15) / 16 + (q + 15) / 16, NG = (m + 15) / 16;
    const size_t tiles = (size_t)(NP * (NP + 1) / 2 > NG * NP ? NP * (NP + 1) / 2 : NG * NP);
    return tiles * 256 <= tsweep_scratch_elems(nbl) ? nbl : 0;
}

template <int NBL> struct TSweep {
    using T = double;
    static_assert(NBL % 2 == 0 && NBL <= 14, "tile rows in pairs");
    static constexpr int NWM = NBL / 2, NW = NWM + 1, NT = 64 * NW, NS = NBL + 1;      // tile waves, waves, threads, slots
    using TM = TileMat<NBL, 8, true>;          // LDS layout of the factorisation, pivot block, operand tiles
    using Pos = typename TM::Pos;
    static constexpr int XS = TM::XS, SS = TM::SS;

    // tile (I, J) of slot s of tile wave w
    static QPX_DEV int slot_i(int w, int s) { return s < NBL - w ? NBL - 1 - w : w; }
    static QPX_DEV int slot_j(int w, int s) { return s < NBL - w ? s : s - (NBL - w); }

    // the sixteen old rows of panel kn -> X (with_s: its pivot block -> S), the identity into X's own block, the next
    // diagonal tile -> S2; from the tiles of this wave
    static QPX_DEV void publish(const Pos& p, int w, const T (&E)[NS][4], T* scr, int kn, bool with_s)
    {
        T* X = scr + TM::kX;
        T* S = scr + TM::kS;
        T* S2 = scr + TM::kS2;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int I = slot_i(w, s), J = slot_j(w, s);
            if (I == kn) {
                if (J < kn) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(p.g + 4 * r) * XS + 16 * J + p.c] = E[s][r];
                } else {
                    if (with_s) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) S[(p.g + 4 * r) * SS + p.c] = E[s][r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(p.g + 4 * r) * XS + 16 * J + p.c] = (p.g + 4 * r == p.c) ? T(1) : T(0);
                }
            } else if (J == kn && I > kn) {
#pragma unroll
                for (int r = 0; r < 4; ++r) X[p.c * XS + 16 * I + p.g + 4 * r] = E[s][r];
            }
            if (I == kn + 1 && J == kn + 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) S2[r * 64 + p.lane] = E[s][r];
            }
        }
    }

    // every tile of this wave: E(I, J) = keep E(I, J) + AT[I]^T BT[J], keep = 0 in the panel's own row and column;
    // slots in groups of three, their MFMA chains interleaved (k-slice outermost)
    static QPX_DEV void update(const Block& blk, const Pos& p, int w, T (&E)[NS][4], const T* scr, int k, T zr)
    {
        const T* BT = scr + TM::kBT;
        const T* AT = scr + TM::kAT;
        constexpr int GS = 3;
#pragma unroll
        for (int s0 = 0; s0 < NS; s0 += GS) {
            T av[GS][4], bv[GS][4];
#pragma unroll
            for (int u = 0; u < GS; ++u) {
                const int s = s0 + u;
                if (s >= NS) continue;
                const int I = slot_i(w, s), J = slot_j(w, s);
                const T keep = (I == k || J == k) ? zr : T(1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    av[u][r] = AT[I * 256 + r * 64 + p.lane];
                    bv[u][r] = BT[J * 256 + r * 64 + p.lane];
                    E[s][r] *= keep;
                }
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    if (s0 + u >= NS) continue;
                    blk.mfma16x16x4(av[u][q4], bv[u][q4], E[s0 + u]);
                }
        }
    }
};

// The cooperative phases (every thread of the workgroup, plain loops over row-major data; tile t of a round at
// stage[t * 256 + row * 16 + column], which is also the accumulator layout [r * 64 + lane]).
template <int NBL>
QPX_DEV void tsweep_body(const Block& blk, const PrefactorArgs<double>& a, int qp, double* lds)
{
    using T = double;
    using SW = TSweep<NBL>;
    using TM = typename SW::TM;
    constexpr int NS = SW::NS;
    const typename TM::Pos g(blk);
    const bool chain = blk.uniform(blk.wave()) == 0;
    int w = blk.uniform(blk.wave()) - 1;                                        // tile wave index
    const int n = a.n, m = a.m, q = a.q;
    const int NQ = (n + 15) / 16, NA = (q + 15) / 16, NG = (m + 15) / 16;      // tile rows per block
    const int NP = NQ + NA;                                                     // pivot tile rows
    const int oA = 16 * NQ;                                                     // first padded row of the A block
    const FacLayout lay = fac_layout(n, m, q, a.images);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const In<T> Qg(a.Q, (size_t)qp * a.sQ, a.io32), Gg(a.G, (size_t)qp * a.sG, a.io32);
    const In<T> Ag(q > 0 ? a.A : nullptr, (size_t)qp * a.sA, a.io32);
    T* rd = lds;
    T* scr = rd + 16 * NBL;
    T* stage = scr;                     // the staging area: the whole scratch region (unused outside the sweep proper)
    T* csum = scr + tsweep_scratch_elems(NBL);      // column sums of G, four partials per column
    const int nt = blk.nt;
#ifdef QPX_PANEL_PROF
    long long tprof[4];
    tprof[0] = clock64();
#endif

    T E[NS][4];
    // ---- round 1: the lower block triangle of [[Q, A^T], [A, 0]] (padded: identity), tile (I, J) at (I (I+1) / 2 + J) * 256
    // (eight independent global loads in flight per thread: one load per trip is a memory round trip per trip)
    constexpr int U = 8;
    const int n1 = NP * (NP + 1) / 2 * 256;
    for (int e0 = blk.tid; e0 < n1; e0 += U * nt) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nt;
            const int t = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            const int J = t - I * (I + 1) / 2, i = 16 * I + rr, j = 16 * J + cc;
            v[u] = T(0);
            if (e < n1) {
                if (i < oA) v[u] = (i < n && j < n) ? Qg[(size_t)i * n + j] : (i == j ? T(1) : T(0));
                else v[u] = (j < n && i - oA < q) ? Ag[(size_t)(i - oA) * n + j] : ((i == j && i - oA >= q) ? T(1) : T(0));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + u * nt < n1) stage[e0 + u * nt] = v[u];
    }
    blk.sync();
    if (!chain) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int I = SW::slot_i(w, s), J = SW::slot_j(w, s);
#pragma unroll
            for (int r = 0; r < 4; ++r) E[s][r] = I < NP ? stage[(I * (I + 1) / 2 + J) * 256 + r * 64 + g.lane] : T(0);
        }
    }
    blk.sync();
    // ---- round 2: G (NG x NQ tiles; the columns of the A block are zero), tile (Ig, J) at (Ig NQ + J) * 256
    const int n2 = NG * NQ * 256;
    for (int e0 = blk.tid; e0 < n2; e0 += U * nt) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * nt;
            const int t = e >> 8, rr = (e >> 4) & 15, cc = e & 15;
            const int Ig = t / NQ, J = t - Ig * NQ, i = 16 * Ig + rr, j = 16 * J + cc;
            v[u] = (e < n2 && i < m && j < n) ? Gg[(size_t)i * n + j] : T(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + u * nt < n2) stage[e0 + u * nt] = v[u];
    }
    blk.sync();
    if (!chain) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int I = SW::slot_i(w, s), J = SW::slot_j(w, s);
            if (I >= NP && I < NP + NG && J < NQ) {
#pragma unroll
                for (int r = 0; r < 4; ++r) E[s][r] = stage[((I - NP) * NQ + J) * 256 + r * 64 + g.lane];
            }
        }
    }
    // || G^T 1 ||: column sums, four partial sums per column added in a fixed order
    for (int t = blk.tid; t < 4 * 16 * NQ; t += nt) {
        const int j = t % (16 * NQ), part = t / (16 * NQ);
        T s = T(0);
        for (int Ig = part; Ig < NG; Ig += 4) {
            const T* col = stage + (Ig * NQ + (j >> 4)) * 256 + (j & 15);
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) s += col[rr * 16];
        }
        csum[t] = s;
    }
    blk.sync();
    if (blk.wave() == 0) {
        T acc = 0;
        for (int j = blk.lane(); j < n; j += kWave) {
            const T cs = (csum[j] + csum[16 * NQ + j]) + (csum[2 * 16 * NQ + j] + csum[3 * 16 * NQ + j]);
            acc = fma_(cs, cs, acc);
        }
        acc = wave_sum(blk, acc);
        if (blk.lane() == 0) F[lay.scal] = sqrt_(acc);
    }
    blk.sync();
#ifdef QPX_PANEL_PROF
    tprof[1] = clock64();
#endif

    // ---- sweep the NP pivot panels (chain-wave form, qpx_tile.h: two barriers per panel)
    T* S = scr + TM::kS;
    T* Wl = scr + TM::kW;
    T* BT = scr + TM::kBT;
    T* AT = scr + TM::kAT;
    T* S2 = scr + TM::kS2;
    T* X = scr + TM::kX;
    T* flag = scr + TM::kFlag;
    auto kmax_of = [&](int k) { return k < NQ ? n - 16 * k : q - 16 * (k - NQ); };
    QPX_LAUNDER_S(w);
    if (!chain) SW::publish(g, w, E, scr, 0, true);
    blk.sync();
    if (chain) TM::pivot_block(blk, g, scr, rd, 0, kmax_of(0), NQ > 0 ? 1 : -1);
    blk.sync();
    int fail = 0;
#pragma unroll 1
    for (int k = 0; k < NP; ++k) {
        const T zr = flag[0];             // 0 unless a pivot broke down
        if (zr != T(0)) { fail = (int)zr; break; }
        const bool la = k + 1 < NP;
        const typename TM::Pos p = g.fresh();
        {
            T wa[4], nrd[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wa[s] = Wl[p.c * TM::SS + p.g + 4 * s];
#pragma unroll
            for (int r = 0; r < 4; ++r) nrd[r] = -rd[16 * k + p.g + 4 * r];
            if (chain) {
                if (la) {                 // b_{k+1}, then the next pivot block brought up to date on its copy
                    T acc[4], bx[4], ao[4], sacc[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[s] = bx[s] = X[(p.g + 4 * s) * TM::XS + 16 * (k + 1) + p.c];
                        sacc[s] = S2[s * 64 + p.lane];
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) blk.mfma16x16x4(wa[s], bx[s], acc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ao[r] = nrd[r] * acc[r];
#pragma unroll
                    for (int s = 0; s < 4; ++s) blk.mfma16x16x4(ao[s], acc[s], sacc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        S[(p.g + 4 * r) * TM::SS + p.c] = sacc[r];
                        BT[(k + 1) * 256 + r * 64 + p.lane] = acc[r];
                        AT[(k + 1) * 256 + r * 64 + p.lane] = ao[r];
                    }
                }
            } else {                      // the other operand tiles: entries w and w + NWM of the list of the J != k + 1
                blk.template prio<0>();
                const int e0 = w, e1 = w + SW::NWM;
                const int J0 = (la && e0 > k) ? e0 + 1 : e0, J1 = (la && e1 > k) ? e1 + 1 : e1;
                TM::operand_pair(blk, p, scr, J0, J1 < NBL ? J1 : -1, wa, nrd);
                blk.template prio<3>();
            }
        }
        blk.sync();
        if (chain) {
            if (la) TM::pivot_block(blk, g.fresh(), scr, rd, 16 * (k + 1), kmax_of(k + 1), k + 1 < NQ ? 1 : -1);
        } else {
            blk.template prio<0>();
            SW::update(blk, g.fresh(), w, E, scr, k, zr);
            if (la) SW::publish(g.fresh(), w, E, scr, k + 1, false);
            blk.template prio<3>();
        }
        blk.sync();
    }
    if (!fail) fail = (int)flag[0];
    blk.sync();
#ifdef QPX_PANEL_PROF
    tprof[2] = clock64();
#endif
    if (fail) {
        for (size_t e = blk.tid; e < lay.total; e += nt) F[e] = T(0);
        if (blk.tid == 0) a.status[qp] = fail == 1 ? QPX_ST_Q_NOT_SPD : QPX_ST_A_RANK;
        return;
    }

    // ---- the blob.  Round 1: the pivot rows' tiles -> LDS -> -K (n x n, both triangles), -N^T (q x n), S_A^-1 (q x q);
    // the G block's own tiles straight from the registers: they are the loop kernels' tile image of R, negated.
    if (!chain) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int I = SW::slot_i(w, s), J = SW::slot_j(w, s);
            if (I < NP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[(I * (I + 1) / 2 + J) * 256 + r * 64 + g.lane] = E[s][r];
            } else if (I < NP + NG && J >= NP) {
                const int Ig = I - NP, Jg = J - NP;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * Ig + g.g + 4 * r, j = 16 * Jg + g.c;
                    F[lay.Rm + (size_t)(Ig * (Ig + 1) / 2 + Jg) * 256 + r * 64 + g.lane] = (i < m && j < m) ? -E[s][r] : T(0);
                }
            }
        }
    }
    // tiles of the R image beyond the G block: zero (the loop kernel pads T with the identity itself)
    for (int I = NG; I < lay.nbt; ++I)
        for (size_t e = blk.tid; e < (size_t)(I + 1) * 256; e += nt) F[lay.Rm + (size_t)(I * (I + 1) / 2) * 256 + e] = T(0);
    blk.sync();
    auto pivot_elem = [&](int i, int j) -> T {                         // element (i, j) of the swept pivot block, padded indices
        const int I = i >> 4, J = j >> 4;
        return I >= J ? stage[(I * (I + 1) / 2 + J) * 256 + (i & 15) * 16 + (j & 15)]
                      : stage[(J * (J + 1) / 2 + I) * 256 + (j & 15) * 16 + (i & 15)];
    };
    // (rows dealt over the threads' high part, columns over the low part: no integer divisions in the loops)
    // (only whole groups of 128 threads take part: 448 threads = 3 groups, the last 64 threads sit these loops out)
    const int ci = blk.tid >> 7, di = nt >> 7, cj = ci < di ? (blk.tid & 127) : (1 << 30);
    if (cj < n) {
        for (int i = ci; i < n; i += di) F[lay.Kneg + (size_t)i * n + cj] = pivot_elem(i, cj);
        for (int i = ci; i < q; i += di) F[lay.NTn + (size_t)i * n + cj] = pivot_elem(oA + i, cj);
    }
    if (cj < q)
        for (int i = ci; i < q; i += di) F[lay.S11i + (size_t)i * q + cj] = pivot_elem(oA + i, oA + cj);
    blk.sync();
    // Round 2: the G rows' tiles left of the G block -> LDS -> M^T (n x m) and W (m x q), negated
    if (!chain) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int I = SW::slot_i(w, s), J = SW::slot_j(w, s);
            if (I >= NP && I < NP + NG && J < NP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[((I - NP) * NP + J) * 256 + r * 64 + g.lane] = E[s][r];
            }
        }
    }
    blk.sync();
    if (cj < m)                                                         // M^T[i][j] = -(G row j, column i)
        for (int i = ci; i < n; i += di)
            F[lay.MT + (size_t)i * m + cj] = -stage[((cj >> 4) * NP + (i >> 4)) * 256 + (cj & 15) * 16 + (i & 15)];
    if (cj < q)                                                         // W[i][j] = -(G row i, column j of the A block)
        for (int i = ci; i < m; i += di)
            F[lay.W + (size_t)i * q + cj] = -stage[((i >> 4) * NP + NQ + (cj >> 4)) * 256 + (i & 15) * 16 + (cj & 15)];
    if (blk.tid == 0) a.status[qp] = 0;
#ifdef QPX_PANEL_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    tprof[3] = clock64();
    if (blk.tid == 0) {
        for (int i = 0; i < 3; ++i) atomicAdd(&qpx_chain_prof[17 + i], (unsigned long long)(tprof[i + 1] - tprof[i]));
        atomicAdd(&qpx_chain_prof[16], 1ull);
    }
#endif
}

}  // namespace qpx
