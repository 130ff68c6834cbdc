// qpx_wave.h -- the wave-per-QP PDIPM kernel (k_ipm_wave): ONE wave64 owns one QP and keeps the
// m x m condensed KKT matrix T = R + diag(s/z) in REGISTERS for the factorisation.
//
// Why: the workgroup version (ipm_body) spends 64 % of its time in a Cholesky whose every column
// costs a workgroup barrier plus a latency-serialised LDS read-modify-write sweep (measured
// 2600 cycles per column at nineq = 100 on MI355X).  Here
//   * the 64 lanes form an 8 x 8 grid (a = lane & 7, b = lane >> 3); lane (a,b) holds the
//     elements (8*li + a, 8*lj + b) of the lower block triangle, li >= lj, as NB(NB+1)/2 registers
//     (NB = ceil(nineq/8); diagonal 8x8 blocks are held in full);
//   * LDL^T, right-looking: when column k is final its UNSCALED entries are published to LDS
//     (packed column-major) and every lane reads back the NB row values (index a) and NB column
//     values (index b) it needs for the rank-1 update of its registers -- one LDS round trip per
//     column and no barrier (a wave's DS traffic is ordered);
//   * look-ahead: the block column holding column k+1 is updated first and column k+1 is
//     published before the bulk of the rank-1 update of step k, so the LDS round trip of the next
//     column overlaps with ~NB^2/2 FMAs;
//   * the published columns ARE the factor the triangular solves use (no second copy);
//   * R is re-read each iteration from the blob in register layout (one coalesced 512-byte
//     load per register, L2-resident); R z' is computed from the registers before the diagonal
//     is added.
// The IPM control flow is the one of ipm_body (see there for the reference line citations).
#pragma once
#include "qpx_kernels.h"

namespace qpx {

constexpr int wave_tri(int x) { return x * (x + 1) / 2; }
// register index of local block (li, lj), li >= lj
constexpr int widx(int li, int lj) { return li * (li + 1) / 2 + lj; }

// elements of the register-layout copy of R in the factor blob for a given NB
QPX_LAYOUT_HD size_t rw_elems(int nb) { return (size_t)(nb * (nb + 1) / 2) * 64; }

// start of column k (its diagonal entry) in the packed column-major lower triangle of order M8
QPX_DEV int wcol_off(int k, int M8) { return k * M8 - (k * (k - 1)) / 2; }

template <class T, int NB> QPX_DEV void wave_load_R(const Block& b, T (&Tr)[wave_tri(NB)], const T* Rw)
{
    const GlobalRows<T> rows(Rw, wave_tri(NB) * kWave, b.lane());
#pragma unroll
    for (int e = 0; e < wave_tri(NB); ++e) Tr[e] = rows.row(e);
}

// vout = R vin for the symmetric matrix held in registers (before the diagonal is added).
// vin / vout: LDS vectors of length 8*NB (pad entries of vin must be finite).
template <class T, int NB>
QPX_DEV void wave_symv(const Block& b, const T (&Tr)[wave_tri(NB)], const T* vin, T* vout)
{
    const int lane = b.lane(), a = lane & 7, bb = lane >> 3;
    // vin is re-read from LDS at every use (broadcast reads are cheap; holding 2*NB more values
    // next to the NB(NB+1)/2 matrix registers would not fit the register file)
    T racc[NB], cacc[NB];
#pragma unroll
    for (int l = 0; l < NB; ++l) {
        racc[l] = T(0);
        cacc[l] = T(0);
    }
#pragma unroll
    for (int li = 0; li < NB; ++li) {
        const T zri = vin[8 * li + a];
#pragma unroll
        for (int lj = 0; lj <= li; ++lj) {
            const T t = Tr[widx(li, lj)];
            racc[li] = fma_(t, vin[8 * lj + bb], racc[li]);
            if (li != lj) cacc[lj] = fma_(t, zri, cacc[lj]);
        }
    }
#pragma unroll
    for (int l = 0; l < NB; ++l) {
        // row part: sum over b (lanes a + 8b); column part: sum over a
        racc[l] += b.shfl_xor(racc[l], 8);
        racc[l] += b.shfl_xor(racc[l], 16);
        racc[l] += b.shfl_xor(racc[l], 32);
        cacc[l] += b.shfl_xor(cacc[l], 1);
        cacc[l] += b.shfl_xor(cacc[l], 2);
        cacc[l] += b.shfl_xor(cacc[l], 4);
    }
    if (a == bb) {
#pragma unroll
        for (int l = 0; l < NB; ++l) vout[8 * l + a] = racc[l] + cacc[l];
    }
}

template <class T, int NB> QPX_DEV void wave_add_diag(const Block& b, T (&Tr)[wave_tri(NB)], const T* vd)
{
    const int lane = b.lane(), a = lane & 7, bb = lane >> 3;
    if (a == bb) {
#pragma unroll
        for (int l = 0; l < NB; ++l) Tr[widx(l, l)] += vd[8 * l + a];
    }
}

// State carried from one column step to the next (software pipeline): the un-scaled entries of
// the NEXT pivot column this lane needs, fetched from LDS while the bulk of the current rank-1
// update executes.
template <class T, int NB> struct LdlCarry {
    T lrow[NB];   // c_ik for rows i = 8*l + a          (l >= block of the column)
    T lcA, lcB;   // c_jk for row j = 8*KB + b and j = 8*(KB+1) + b
    T dk;         // pivot d_k
};

template <class T, int NB, int KB>
QPX_DEV void wave_ldl_fetch(const T* Lc, int kn, int a, int bb, LdlCarry<T, NB>& c)
{
    // column kn (belonging to block KB or starting block KB): rows are read from block KB on
    constexpr int M8 = 8 * NB;
    const int offn = wcol_off(kn, M8);
    const int base = offn - kn;
#pragma unroll
    for (int l = KB; l < NB; ++l) c.lrow[l] = Lc[base + 8 * l + a];
    c.lcA = Lc[base + 8 * KB + bb];
    c.lcB = (KB + 1 < NB) ? Lc[base + 8 * (KB + 1) + bb] : T(0);
    c.dk = Lc[offn];
}

// One column step (column k = 8*KB + ka); KBN = block column of column k+1 (KB, or KB+1 when
// ka == 7).  `c` holds column k on entry and column k+1 on exit.  Returns false on a
// non-positive / non-finite pivot.
template <class T, int NB, int KB, int KBN>
QPX_DEV bool wave_ldl_step(const Block& b, T (&Tr)[wave_tri(NB)], T* Lc, T* rd, int ka, LdlCarry<T, NB>& c)
{
    constexpr int M8 = 8 * NB;
    const int lane = b.lane(), a = lane & 7, bb = lane >> 3;
    const int k = 8 * KB + ka;
    const int base = wcol_off(k, M8) - k;          // row i of column k sits at base + i  (i >= k)
    const T dk = c.dk;
    if (!(dk > T(0)) || !finite_(dk)) return false;
    const T r = rcp_(dk);
    if (lane == 0) rd[k] = r;
    T lrs[NB + 1];
#pragma unroll
    for (int l = KBN; l < NB; ++l) lrs[l] = c.lrow[l] * r;
    if constexpr (KBN < NB) {
        // look-ahead: finish the block column that holds column k+1 and publish that column
        const T lc0 = (KBN == KB) ? c.lcA : c.lcB;
#pragma unroll
        for (int li = KBN; li < NB; ++li) Tr[widx(li, KBN)] = fma_(-lrs[li], lc0, Tr[widx(li, KBN)]);
        const int kn = k + 1;
        const int kan = kn & 7;
        const int offn = wcol_off(kn, M8) - kn;
        if (bb == kan) {
#pragma unroll
            for (int li = KBN; li < NB; ++li) {
                const int i = 8 * li + a;
                if (i >= kn) Lc[offn + i] = Tr[widx(li, KBN)];
            }
        }
        b.wave_sync();
        // software pipeline: the reads of column k+1 go out now and land during the bulk update
        wave_ldl_fetch<T, NB, KBN>(Lc, kn, a, bb, c);
        // bulk of the rank-1 update with column k
#pragma unroll
        for (int lj = KBN + 1; lj < NB; ++lj) {
            const T lc = Lc[base + 8 * lj + bb];
#pragma unroll
            for (int li = lj; li < NB; ++li) Tr[widx(li, lj)] = fma_(-lrs[li], lc, Tr[widx(li, lj)]);
        }
    }
    return true;
}

template <class T, int NB, int KB> struct WaveLdlBlocks {
    static QPX_DEV bool run(const Block& b, T (&Tr)[wave_tri(NB)], T* Lc, T* rd, LdlCarry<T, NB>& c)
    {
#pragma unroll 1
        for (int ka = 0; ka < 7; ++ka)
            if (!wave_ldl_step<T, NB, KB, KB>(b, Tr, Lc, rd, ka, c)) return false;
        if (!wave_ldl_step<T, NB, KB, KB + 1>(b, Tr, Lc, rd, 7, c)) return false;
        return WaveLdlBlocks<T, NB, KB + 1>::run(b, Tr, Lc, rd, c);
    }
};
template <class T, int NB> struct WaveLdlBlocks<T, NB, NB> {
    static QPX_DEV bool run(const Block&, T (&)[wave_tri(NB)], T*, T*, LdlCarry<T, NB>&) { return true; }
};

// T = L~ D L~^T.  On return Lc holds the UNIT lower factor packed by columns (l~_ik below the
// diagonal; the diagonal slots keep d_k) and rd[k] = 1/d_k.  The registers are consumed.
template <class T, int NB>
QPX_DEV bool wave_ldl(const Block& b, T (&Tr)[wave_tri(NB)], T* Lc, T* rd, int m)
{
    constexpr int M8 = 8 * NB;
    const int lane = b.lane(), a = lane & 7, bb = lane >> 3;
    if (bb == 0) {
#pragma unroll
        for (int li = 0; li < NB; ++li) Lc[8 * li + a] = Tr[widx(li, 0)];      // column 0
    }
    b.wave_sync();
    LdlCarry<T, NB> c;
    wave_ldl_fetch<T, NB, 0>(Lc, 0, a, bb, c);
    if (!WaveLdlBlocks<T, NB, 0>::run(b, Tr, Lc, rd, c)) return false;
    b.wave_sync();
    // scale column k by 1/d_k: the substitutions then carry no multiply in their dependent chain
    for (int k = 0; k + 1 < m; ++k) {
        const T r = rd[k];
        const int off = wcol_off(k, M8) - k;
        for (int i = k + 1 + lane; i < m; i += kWave) Lc[off + i] *= r;
    }
    b.wave_sync();
    return true;
}

// Solve L~ D u = x in place (x -> u), L~ unit lower (column k contiguous), vector of length m in
// slot layout (element i in slot i/64 of lane i%64).  Dependent chain per step: readlane -> fma.
// The column entries of the next UNR steps are fetched while the current UNR steps execute.
template <int NS, class T>
QPX_DEV void wtrsv_fwd(const Block& b, const T* Lc, const T* rd, int M8, int m, T (&x)[NS])
{
    constexpr int UNR = 4;
    const int lane = b.lane();
    const int ngroups = (m + UNR - 1) / UNR;
    T cur[UNR][NS], nxt[UNR][NS];
    auto fetch = [&](int g, T (&dst)[UNR][NS]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = g * UNR + u;
            const int off = wcol_off(k, M8) - k;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const int i = s2 * kWave + lane;
                const bool v = (k < m) && (i > k) && (i < m);
                const T val = Lc[v ? off + i : 0];
                dst[u][s2] = v ? val : T(0);
            }
        }
    };
    fetch(0, cur);
    for (int g = 0; g < ngroups; ++g) {
        if (g + 1 < ngroups) fetch(g + 1, nxt);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = g * UNR + u;
            if (k < m) {
                const int sk = k >> 6, lk = k & 63;
                T yk = T(0);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (s == sk) yk = b.bcast(x[s], lk);
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) x[s2] = fma_(-cur[u][s2], yk, x[s2]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) cur[u][s2] = nxt[u][s2];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = s * kWave + lane;
        x[s] = (i < m) ? x[s] * rd[i] : T(0);
    }
}

// Solve L~^T y = u in place.  Row k of the packed factor is read per step (lane j reads l~_kj).
template <int NS, class T>
QPX_DEV void wtrsv_bwd(const Block& b, const T* Lc, const T* rd, int M8, int m, T (&x)[NS])
{
    constexpr int UNR = 4;
    const int lane = b.lane();
    const int ngroups = (m + UNR - 1) / UNR;
    int offj[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = s * kWave + lane;
        offj[s] = (j < m) ? (wcol_off(j, M8) - j) : 0;     // l~_kj sits at offj + k  (k > j)
    }
    T cur[UNR][NS], nxt[UNR][NS];
    auto fetch = [&](int g, T (&dst)[UNR][NS]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = m - 1 - (g * UNR + u);
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const int j = s2 * kWave + lane;
                const bool v = (k >= 0) && (j < k);
                const T val = Lc[v ? offj[s2] + k : 0];
                dst[u][s2] = v ? val : T(0);
            }
        }
    };
    fetch(0, cur);
    for (int g = 0; g < ngroups; ++g) {
        if (g + 1 < ngroups) fetch(g + 1, nxt);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = m - 1 - (g * UNR + u);
            if (k >= 0) {
                const int sk = k >> 6, lk = k & 63;
                T yk = T(0);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (s == sk) yk = b.bcast(x[s], lk);
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) x[s2] = fma_(-cur[u][s2], yk, x[s2]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) cur[u][s2] = nxt[u][s2];
    }
    (void)rd;
}

// dz = -T^-1 rhs with T = L~ D L~^T
template <int NS, class T>
QPX_DEV void wave_solve_neg(const Block& b, const T* Lc, const T* rd, int M8, int m, T (&x)[NS])
{
    wtrsv_fwd<NS>(b, Lc, rd, M8, m, x);
    wtrsv_bwd<NS>(b, Lc, rd, M8, m, x);
#pragma unroll
    for (int s = 0; s < NS; ++s) x[s] = -x[s];
}

QPX_LAYOUT_HD size_t lds_elems_ipm_wave(int n, int nb, int q)
{
    const size_t m8 = (size_t)8 * nb;
    const size_t v = align4(max2(max2((size_t)n, m8), (size_t)q));
    return align4(max2(tri(m8), tri((size_t)n))) + 12 * v + 4;
}

// ------------------------------------------------------------------------------------------
// The PDIPM loop, one wave per QP.  Same mathematics and control flow as ipm_body.
template <class T, int NB, int NS>
QPX_DEV void ipm_wave_body(const Block& b, const IpmArgs<T>& a, int qp, T* lds)
{
    constexpr int M8 = 8 * NB;
    const int n = a.n, m = a.m, q = a.q;
    const FacLayout lay = fac_layout(n, m, q);
    T* F = a.fac + (size_t)qp * a.fac_stride;
    const T* Rw = F + lay.Rw;
    const size_t v = align4(max2(max2((size_t)n, (size_t)M8), (size_t)q));
    T* dinv = lds;        // 1/L_kk of Q's factor (n)
    T* rd = dinv + v;     // 1/d_k of the current T factor (M8)
    T* vA = rd + v;       // z' (M8)
    T* vB = vA + v;       // R z' (M8) ; later w (n)
    T* vC = vB + v;       // c (m)
    T* vW = vC + v;       // w0 (n)
    T* vT = vW + v;       // t = L^-1 p (n)
    T* vQ1 = vT + v;      // ycoef (q)
    T* vQ2 = vQ1 + v;     // beta (q)
    T* vD = vQ2 + v;      // s/z, 1 on the pad (M8)
    T* vBZ = vD + v;      // best z (m)
    T* vBS = vBZ + v;     // best s (m)
    T* Lc = vBS + v + 4;  // packed factor of T (tri(M8)) / packed L of Q (tri(n))

    const int lane = b.lane();
    const T mT = (T)m;
    const T* pg = a.p + (size_t)qp * a.sp;
    const T* hg = a.h + (size_t)qp * a.sh;
    const T* bg = q > 0 ? a.b + (size_t)qp * a.sb : nullptr;

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

    QPX_PROF_INIT
    // ---- constants that depend on p, h, b (see ipm_body)
    {
        const T* Zp = F + lay.Zp;
        const T* Yh = F + lay.Yh;
        const T* V = F + lay.V;
        block_copy(b, Lc, F + lay.L, (size_t)tri(n));
        block_copy(b, dinv, F + lay.dinvL, (size_t)n);
        b.sync();
        {
            T x[NS];
            ld_slots<NS>(b, x, pg, n, T(0));
            trsv_fwd<NS>(b, Lc, dinv, n, x);
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
        for (int j = b.tid; j < M8; j += b.nt) {
            T acc = T(0);
            if (j < m) {
                acc = hg[j];
                for (int k = 0; k < n; ++k) acc = fma_(Zp[(size_t)k * m + j], vT[k], acc);
                for (int r = 0; r < q; ++r) acc = fma_(-V[(size_t)r * m + j], vQ2[r], acc);
            }
            vC[j] = acc;
            vD[j] = T(1);
            vA[j] = T(0);
        }
        b.sync();
        for (int r = b.tid; r < q; r += b.nt) vQ1[r] += vQ2[r];
    }
    QPX_PROF(0)

    T Tr[wave_tri(NB)];
    T z[NS], s[NS];
    T tau = 1, btau = 1, sigz = 0, sigs = 0, bres = Lim<T>::inf();
    const T g1n = F[lay.scal];
    T feas_prev = 0, alpha_prev = 0;
    int nnot = 0, floor_hit = 0, st = 0, iters = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) z[k] = s[k] = T(1);

    // ---- start point: T = R + I, z_i = -T^-1 c, s_i = -z_i, shifts (batch.py:61-87)
    wave_load_R<T, NB>(b, Tr, Rw);
    wave_add_diag<T, NB>(b, Tr, vD);
    bool ok = wave_ldl<T, NB>(b, Tr, Lc, rd, m);
    int stop = 0;
    if (ok) {
        T x[NS];
        ld_slots<NS>(b, x, vC, m, T(0));
        wave_solve_neg<NS>(b, Lc, rd, M8, m, x);
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
        sigz = (mnz < T(0)) ? (T(1) - mnz) : T(0);
        sigs = (mns < T(0)) ? (T(1) - mns) : T(0);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) {
                z[k] = x[k] + sigz;
                s[k] = -x[k] + sigs;
                vA[i] = x[k];
                vBZ[i] = z[k];
                vBS[i] = s[k];
            }
        }
    } else {
        st |= QPX_ST_KKT_BREAKDOWN;
        stop = 1;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) { vBZ[i] = T(1); vBS[i] = T(1); }
        }
    }
    b.wave_sync();
    QPX_PROF(1)

    for (int it = 0; it < a.maxIter && !stop; ++it) {
        wave_load_R<T, NB>(b, Tr, Rw);
        QPX_PROF(2)
        wave_symv<T, NB>(b, Tr, vA, vB);
        b.wave_sync();
        QPX_PROF(3)
        T pri2 = 0, szdot = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = k * kWave + lane;
            if (i < m) {
                const T rz = s[k] - vC[i] - vB[i];
                pri2 = fma_(rz, rz, pri2);
                szdot = fma_(s[k], z[k], szdot);
                vD[i] = s[k] / z[k];
            }
        }
        pri2 = wave_sum(b, pri2);
        szdot = wave_sum(b, szdot);
        const T mu = abs_(szdot / mT);
        const T pri = sqrt_(pri2);
        const T dual = tau * sigz * g1n;
        const T feas = pri + dual;
        const T resid = feas + mT * mu;
        if (a.trace && lane == 0) {
            T* tr = a.trace + ((size_t)it * a.B + qp) * 3;
            tr[0] = pri; tr[1] = dual; tr[2] = mu;
        }
        b.wave_sync();
        wave_add_diag<T, NB>(b, Tr, vD);
        QPX_PROF(4)
        ok = wave_ldl<T, NB>(b, Tr, Lc, rd, m);
        QPX_PROF(5)
        int stopf = 0;
        if (!ok) {
            st |= QPX_ST_KKT_BREAKDOWN;
            stopf = 1;
        } else {
            iters = it + 1;
            const bool better = (it == 0) || (resid < bres);
            if (better) {
                bres = resid; btau = tau; nnot = 0;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int i = k * kWave + lane;
                    if (i < m) { vBZ[i] = z[k]; vBS[i] = s[k]; }
                }
            } else if (a.stall_policy == 1 || (a.stall_policy == 2 && mT * mu < feas)) {
                nnot += 1;
            } else {
                nnot = 0;
            }
            if (a.stall_policy == 2 && it >= 1 && feas > T(2) * (T(1) - alpha_prev) * feas_prev) floor_hit = 1;
            feas_prev = feas;
            if ((a.stall_policy != 0 && nnot >= a.notImprovedLim) || bres < a.eps || mu > T(1e32)) stopf = 1;
            if (a.stall_policy == 2 && floor_hit && mT * mu < T(1e-2) * feas) stopf = 1;
            if (!finite_(resid)) { stopf = 1; st |= QPX_ST_NONFINITE; }
        }
        if (!stopf) {
            T dza[NS], dsa[NS], dz[NS], ds[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                dza[k] = (i < m) ? (vC[i] + vB[i] + tau * sigz * F[lay.r1 + i]) : T(0);
            }
            wave_solve_neg<NS>(b, Lc, rd, M8, m, dza);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                dsa[k] = (i < m) ? (-s[k] - dza[k] * s[k] / z[k]) : T(0);
                if (i >= m) dza[k] = T(0);
            }
            T al = step_to_boundary<NS>(b, z, dza, m);
            const T al2 = step_to_boundary<NS>(b, s, dsa, m);
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
            T rs[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                rs[k] = (i < m) ? ((-mu * sig + dsa[k] * dza[k]) / s[k]) : T(0);
                dz[k] = (i < m) ? (rs[k] * s[k] / z[k]) : T(0);
            }
            wave_solve_neg<NS>(b, Lc, rd, M8, m, dz);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                const T dsc = (i < m) ? ((-rs[k] - dz[k]) * s[k] / z[k]) : T(0);
                dz[k] = (i < m) ? (dza[k] + dz[k]) : T(0);
                ds[k] = dsa[k] + dsc;
            }
            al = step_to_boundary<NS>(b, z, dz, m);
            const T al3 = step_to_boundary<NS>(b, s, ds, m);
            al = (al3 < al) ? al3 : al;
            al = T(0.999) * al;
            al = (al < T(1)) ? al : T(1);
            tau = (T(1) - al) * tau;
            alpha_prev = al;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int i = k * kWave + lane;
                if (i < m) {
                    z[k] = fma_(al, dz[k], z[k]);
                    s[k] = fma_(al, ds[k], s[k]);
                    vA[i] = z[k] - tau * sigz;
                }
            }
        }
        stop = stopf;
        b.wave_sync();
        QPX_PROF(6)
    }

    // ---- outputs (batch.py:143,207)
    if (iters >= a.maxIter && !(bres < a.eps)) st |= QPX_ST_MAXITER;
    if (!(bres <= T(1))) st |= QPX_ST_INACCURATE;
    for (int i = lane; i < m; i += kWave) {
        const T bz = vBZ[i];
        a.lam[(size_t)qp * m + i] = bz;
        a.slack[(size_t)qp * m + i] = vBS[i];
        vA[i] = bz - btau * sigz;
    }
    if (lane == 0) {
        a.iters[qp] = iters;
        a.status[qp] |= st;
        a.best_resid[qp] = bres;
    }
    b.sync();
    {
        const T* Zp = F + lay.Zp;
        // w = w0 - Zp z' : four rows per pass so the wave reductions overlap
        for (int k0 = 0; k0 < n; k0 += 4) {
            T acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = T(0);
                if (k0 + u < n) {
                    const T* zr = Zp + (size_t)(k0 + u) * m;
                    for (int j = lane; j < m; j += kWave) acc[u] = fma_(zr[j], vA[j], acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = wave_sum(b, acc[u]);
                if (lane == 0 && k0 + u < n) vB[k0 + u] = vW[k0 + u] - acc[u];
            }
        }
    }
    block_copy(b, Lc, F + lay.L, (size_t)tri(n));
    b.sync();
    {
        T x[NS];
        ld_slots<NS>(b, x, vB, n, T(0));
        trsv_bwd<NS>(b, Lc, dinv, n, x);
        st_slots<NS>(b, a.zhat + (size_t)qp * n, x, n);
        if (q > 0) {
            const T* V = F + lay.V;
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
                        if (lane == la) y[sa] = vQ1[r] + acc;
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

}  // namespace qpx
