// qpx_big_polish.h -- the FINISHING STAGE of the large-QP family (round 5; qpx_polish beyond nz + neq + nineq = 208).
//
// `steps` iterations of the reference's PDIPM loop in the ORIGINAL variables (qpth/solvers/pdipm/batch.py:92-198: affine +
// centring-corrector Newton steps, step lengths batch.py:189-198, best iterate batch.py:118-139) started from the loop's
// result, with the KKT residuals (batch.py:93-101) formed from the caller's Q, p, G, h, A, b in float64 accumulation whatever
// the element type -- what KKTSolvers.IR_UNOPT and QPFunction(refine=k) on float32 tensors ask for.  Rounds 3-4 ran this
// stage for the large sizes as ~30 eager torch operations per step on the host side (KKTFactors._polish_host); here it
// is a stream-ordered sequence of launches behind the C ABI like everything else of the family:
//
//   per step:  stage 1  residuals of the iterate, mu, the reference's total residual, best iterate kept (strict <, NaN
//                       never wins); d = z/s and the affine right-hand side into the blob's solve vectors
//              the family's KKT solve (factorisation of T = R + diag(s/z) + the condensed solve: qpx_api.inc, big_kkt_core)
//              stage 2  affine direction -> step length, sigma, the corrector's right-hand side
//              the KKT solve again on the SAME factor
//              stage 3  full step, step length (0.999 damping), update of the iterate
//   at the end: stage 1 once more (residuals of the last iterate, best iterate) and the outputs.
//
// One workgroup of sixteen waves per QP.  The iterate and the best iterate live in DOUBLE in the blob (BigLayout::pol)
// whatever T is: in float32 the rounding of x alone puts a floor of ~eps32 ||Q|| ||x|| under the residual that ranks the
// iterates (qpx_grid.h: polish_mat_role, which this follows step by step).  The mat-vecs with the caller's matrices read
// every matrix ONCE per evaluation: a wave owns rows w, w + 16, ... and forms the row dot (G x, A x, Q x) by a lane
// reduction and, from the same loaded row, its contribution to the column sums (G^T z, A^T y) in per-lane accumulators
// that meet in LDS.
#pragma once
#include "qpx_big.h"

namespace qpx {

template <class T> struct BigPolishArgs {
    int B, n, m, q;
    int stage;                            // 1, 2, 3 as above
    int first, last;                      // stage 1: load the iterate from the caller's arrays first / write the outputs and stop
    T* fac; size_t fac_stride;
    const T *Q, *G, *A; long long sQ, sG, sA;
    const T *p, *h, *b; long long sp, sh, sb;
    T *zhat, *nu, *lam, *slack;           // in: the iterate to start from; out: the best iterate met
    T* best_resid;                        // may be null
    int* status;
};
constexpr int kBigPolWaves = 16;
// LDS (doubles): x, s, z, y, rx, rz, ry (7 vectors of VP) + sixteen column partial sums of VP + 16 scalars
QPX_LAYOUT_HD size_t big_polish_lds_doubles(int vp) { return (size_t)(7 + kBigPolWaves) * vp + 16; }

enum BigPolScal { bpMu = 0, bpTot, bpBest, bpDead, bpAlpha, bpBetter };

template <class T> QPX_DEV void big_polish_body(const Block& b, const BigPolishArgs<T>& a, int qp, double* lds)
{
    const BigLayout L = big_layout(a.n, a.m, a.q);
    const int n = a.n, m = a.m, q = a.q, VP = L.VP;
    T* F = a.fac + (size_t)qp * a.fac_stride;
    int* ctrl = reinterpret_cast<int*>(F + L.ctrl);
    double* P = reinterpret_cast<double*>(F + L.pol);
    double *xd = P, *sd = P + VP, *zd = P + 2 * VP, *yd = P + 3 * VP;
    double *bxd = P + 4 * VP, *bsd = P + 5 * VP, *bzd = P + 6 * VP, *byd = P + 7 * VP;
    double* sc = P + 8 * VP;              // 16 scalars
    T *vD = F + L.v(bvD), *vRH = F + L.v(bvRH), *vU = F + L.v(bvU), *vX = F + L.v(bvX), *vW = F + L.v(bvW);
    T *vBQ = F + L.v(bvBQ), *vNU = F + L.v(bvNU), *vTB = F + L.v(bvTB), *vT1 = F + L.v(bvT1);
    // the loop's vectors are dead once qpx_ipm has returned its outputs: the affine direction and the corrector's
    // right-hand side are kept in five of them
    T *pDZA = F + L.v(bvDZA), *pDSA = F + L.v(bvDSA), *pDXA = F + L.v(bvP), *pDYA = F + L.v(bvC), *pRSC = F + L.v(bvRSC);
    double* xl = lds;                     // the iterate (stage 1) / work vectors (stages 2, 3)
    double* sl = xl + VP;
    double* zl = sl + VP;
    double* yl = zl + VP;
    double* rxl = yl + VP;
    double* rzl = rxl + VP;
    double* ryl = rzl + VP;
    double* part = ryl + VP;              // kBigPolWaves x VP
    double* scl = part + (size_t)kBigPolWaves * VP;
    const int lane = b.lane(), w = b.uniform(b.wave()), nw = b.nwaves(), NT = b.nt;
    const double mD = (double)m, tiny = (double)Lim<T>::tiny();
    const int pre_bits = QPX_ST_Q_NOT_SPD | QPX_ST_A_RANK;

    // min over dv < 0 of -v / dv (inf if none), by one wave
    auto step_len = [&](const double* vv, const double* dv) {
        double al = __builtin_huge_val();
        for (int i = lane; i < m; i += kWave) {
            const double d = dv[i];
            if (d < 0.0) al = min2_(al, -vv[i] / d);
        }
        return wave_min(b, al);
    };
    // the solve vectors of a KKT solve with right-hand sides (rx, rs, rz, ry) = (ux, rs, rz, uy): vD = s/z is in place
    auto zero_solve_work = [&](int i) { vW[i] = T(0); vTB[i] = T(0); vT1[i] = T(0); vNU[i] = T(0); };

    if (a.stage == 1) {
        if (a.first) {
            for (int i = b.tid; i < VP; i += NT) {
                const double x = (i < n) ? (double)a.zhat[(size_t)qp * n + i] : 0.0;
                const double s = (i < m) ? (double)a.slack[(size_t)qp * m + i] : 1.0;
                const double z = (i < m) ? (double)a.lam[(size_t)qp * m + i] : 1.0;
                const double y = (i < q) ? (double)a.nu[(size_t)qp * q + i] : 0.0;
                xd[i] = bxd[i] = xl[i] = x;
                sd[i] = bsd[i] = sl[i] = s;
                zd[i] = bzd[i] = zl[i] = z;
                yd[i] = byd[i] = yl[i] = y;
            }
            if (b.tid == 0) { sc[bpBest] = __builtin_huge_val(); sc[bpDead] = 0.0; }
        } else {
            for (int i = b.tid; i < VP; i += NT) { xl[i] = xd[i]; sl[i] = sd[i]; zl[i] = zd[i]; yl[i] = yd[i]; }
        }
        if (b.tid == 0) { scl[bpBest] = a.first ? __builtin_huge_val() : sc[bpBest]; scl[bpDead] = a.first ? 0.0 : sc[bpDead]; }
        b.sync();
        // ---- rx = Q x + p + G^T z + A^T y,  rz = G x + s - h,  ry = A x - b     (batch.py:93-101), double accumulation
        const int nk = (n + kWave - 1) / kWave;
        int ck[8];
        double xk[8], cacc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = lane + kWave * k;
            ck[k] = c < n ? c : n - 1;                     // clamped: every lane loads, the value is deselected
            xk[k] = (k < nk && c < n) ? xl[c] : 0.0;
            cacc[k] = 0.0;
        }
        const T* Qg = a.Q + (size_t)qp * a.sQ;
        const T* Gg = a.G + (size_t)qp * a.sG;
        const T* Ag = q > 0 ? a.A + (size_t)qp * a.sA : nullptr;
        const T* pg = a.p + (size_t)qp * a.sp;
        const T* hg = a.h + (size_t)qp * a.sh;
        const T* bg = q > 0 ? a.b + (size_t)qp * a.sb : nullptr;
        // one row of a matrix: the row dot with x (returned, summed over the lanes) and `coef` times the row into cacc
        auto row = [&](const T* M, int r, double coef, bool cols) {
            const T* Mr = M + (size_t)r * n;
            double v[8], dot = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nk) v[k] = (double)Mr[ck[k]];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nk) {
                    dot = fma_(v[k], xk[k], dot);
                    if (cols) cacc[k] = fma_(coef, v[k], cacc[k]);
                }
            return wave_sum(b, dot);
        };
        for (int r = w; r < n; r += nw) {
            const double d = row(Qg, r, 0.0, false);
            if (lane == 0) rxl[r] = d + (double)pg[r];
        }
        for (int r = w; r < m; r += nw) {
            const double d = row(Gg, r, zl[r], true);
            if (lane == 0) rzl[r] = d + sl[r] - (double)hg[r];
        }
        for (int r = w; r < q; r += nw) {
            const double d = row(Ag, r, yl[r], true);
            if (lane == 0) ryl[r] = d - (double)bg[r];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nk) part[(size_t)w * VP + lane + kWave * k] = cacc[k];
        b.sync();
        for (int c = b.tid; c < n; c += NT) {
            double sum = rxl[c];
            for (int ww = 0; ww < nw; ++ww) sum += part[(size_t)ww * VP + c];
            rxl[c] = sum;
        }
        b.sync();
        if (w == 0) {
            double sz = 0, nx = 0, nz = 0, ny = 0;
            for (int i = lane; i < m; i += kWave) { sz = fma_(sl[i], zl[i], sz); nz = fma_(rzl[i], rzl[i], nz); }
            for (int i = lane; i < n; i += kWave) nx = fma_(rxl[i], rxl[i], nx);
            for (int i = lane; i < q; i += kWave) ny = fma_(ryl[i], ryl[i], ny);
            sz = wave_sum(b, sz); nx = wave_sum(b, nx); nz = wave_sum(b, nz); ny = wave_sum(b, ny);
            const double mu = abs_(sz) / mD;
            const double tot = sqrt_(nx) + sqrt_(nz) + sqrt_(ny) + mD * mu;            // batch.py:103-107
            const bool better = tot < scl[bpBest];                   // false for NaN: a non-finite iterate never wins
            b.wave_sync();
            if (lane == 0) {
                scl[bpMu] = mu; scl[bpBetter] = better ? 1.0 : 0.0;
                sc[bpMu] = mu; sc[bpTot] = tot;
                if (better) { sc[bpBest] = tot; scl[bpBest] = tot; }
            }
        }
        b.sync();
        if (scl[bpBetter] != 0.0 && !a.first) {
            for (int i = b.tid; i < VP; i += NT) { bxd[i] = xl[i]; bsd[i] = sl[i]; bzd[i] = zl[i]; byd[i] = yl[i]; }
        }
        if (a.last) {
            // (the best iterate as this workgroup knows it: from LDS if it has just won, else from the blob)
            const bool now = scl[bpBetter] != 0.0 && !a.first;
            for (int i = b.tid; i < n; i += NT) a.zhat[(size_t)qp * n + i] = (T)(now || a.first ? xl[i] : bxd[i]);
            for (int i = b.tid; i < m; i += NT) {
                a.lam[(size_t)qp * m + i] = (T)(now || a.first ? zl[i] : bzd[i]);
                a.slack[(size_t)qp * m + i] = (T)(now || a.first ? sl[i] : bsd[i]);
            }
            for (int i = b.tid; i < q; i += NT) a.nu[(size_t)qp * q + i] = (T)(now || a.first ? yl[i] : byd[i]);
            if (b.tid == 0) {
                if (a.best_resid) a.best_resid[qp] = (T)scl[bpBest];
                if (scl[bpDead] != 0.0 && a.status) a.status[qp] |= QPX_ST_KKT_BREAKDOWN;
            }
            return;
        }
        if (scl[bpDead] != 0.0) return;
        // ---- the affine solve's inputs: solve_kkt(rx, rs = z, rz, ry) with d = z/s   (batch.py:146,160-162)
        for (int i = b.tid; i < VP; i += NT) {
            double d = 1.0;
            if (i < m) d = max2_(sl[i], tiny) / max2_(zl[i], tiny);              // s/z = 1/d, clamped as the host version did
            const T dT = (T)d;
            vD[i] = dT;
            vRH[i] = (i < m) ? (T)zl[i] * dT - (T)rzl[i] : T(0);
            vU[i] = (i < n) ? (T)rxl[i] : T(0);
            vBQ[i] = (i < q) ? (T)ryl[i] : T(0);
            zero_solve_work(i);
        }
        if (b.tid == 0) { ctrl[bcStop] = 0; ctrl[bcFail] &= pre_bits; }
        return;
    }

    if (sc[bpDead] != 0.0) return;
    if (ctrl[bcFail] & QPX_ST_KKT_BREAKDOWN) {           // the factorisation of this step broke down: the best iterate stands
        if (b.tid == 0) sc[bpDead] = 1.0;
        return;
    }
    double* dza = xl;                     // stage 2 / 3 work vectors in LDS (doubles of the T-rounded directions)
    double* dsa = sl;
    double* dzf = zl;
    double* dsf = yl;
    if (a.stage == 2) {
        for (int i = b.tid; i < VP; i += NT) {
            const T zT = (i < m) ? (T)zd[i] : T(0);
            const T dz = (i < m) ? vX[i] : T(0);
            const T ds = (i < m) ? (-zT - dz) * vD[i] : T(0);
            pDZA[i] = dz; pDSA[i] = ds;
            pDXA[i] = (i < n) ? vW[i] : T(0);
            pDYA[i] = (i < q) ? vNU[i] : T(0);
            dza[i] = (double)dz; dsa[i] = (double)ds;
            rxl[i] = (i < m) ? sd[i] : 1.0;
            rzl[i] = (i < m) ? zd[i] : 1.0;
        }
        b.sync();
        if (w == 0) {
            double al = min2_(step_len(rzl, dza), step_len(rxl, dsa));
            al = min2_(al, 1.0);
            double t3 = 0, sz = 0;
            for (int i = lane; i < m; i += kWave) {
                t3 = fma_(rxl[i] + al * dsa[i], rzl[i] + al * dza[i], t3);
                sz = fma_(rxl[i], rzl[i], sz);
            }
            t3 = wave_sum(b, t3); sz = wave_sum(b, sz);
            double sig = t3 / sz;
            sig = sig * sig * sig;                                   // batch.py:168
            const double mu = sc[bpMu];
            for (int i = lane; i < m; i += kWave)
                pRSC[i] = (T)((-mu * sig + dsa[i] * dza[i]) / max2_(rxl[i], tiny));       // batch.py:171
            for (int i = lane + m; i < VP; i += kWave) pRSC[i] = T(0);
        }
        b.sync();
        // ---- the corrector solve's inputs: solve_kkt(0, rs_cor, 0, 0)
        for (int i = b.tid; i < VP; i += NT) {
            vRH[i] = (i < m) ? pRSC[i] * vD[i] : T(0);
            vU[i] = T(0);
            vBQ[i] = T(0);
            zero_solve_work(i);
        }
        return;
    }
    // stage 3: the full step, its length, the new iterate
    for (int i = b.tid; i < VP; i += NT) {
        const T dzc = (i < m) ? vX[i] : T(0);
        const T dsc = (i < m) ? (-pRSC[i] - dzc) * vD[i] : T(0);
        dzf[i] = (i < m) ? (double)(T)(pDZA[i] + dzc) : 0.0;
        dsf[i] = (i < m) ? (double)(T)(pDSA[i] + dsc) : 0.0;
        rxl[i] = (i < m) ? sd[i] : 1.0;
        rzl[i] = (i < m) ? zd[i] : 1.0;
    }
    b.sync();
    if (w == 0) {
        double al = 0.999 * min2_(step_len(rzl, dzf), step_len(rxl, dsf));                      // batch.py:193
        al = min2_(al, 1.0);
        b.wave_sync();
        if (lane == 0) scl[bpAlpha] = al;
    }
    b.sync();
    {
        const double al = scl[bpAlpha];
        for (int i = b.tid; i < VP; i += NT) {
            if (i < n) xd[i] = fma_(al, (double)(T)(pDXA[i] + vW[i]), xd[i]);
            if (i < m) { sd[i] = fma_(al, dsf[i], rxl[i]); zd[i] = fma_(al, dzf[i], rzl[i]); }
            if (i < q) yd[i] = fma_(al, (double)(T)(pDYA[i] + vNU[i]), yd[i]);
        }
    }
}

}  // namespace qpx
