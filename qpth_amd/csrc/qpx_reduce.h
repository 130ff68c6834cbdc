// qpx_reduce.h -- batch-mean gradient of a SHARED parameter as one contraction over the batch.
//
// The reference forms B outer products per shared parameter and averages them (qp.py:159-177:
// `dQs = bger(dx, zhat) ...; dQs = dQs.mean(0)`), i.e. it writes and re-reads B x n x n numbers to
// produce n x n.  Summed over the batch the outer products are a dense matrix product with the
// batch as the contraction index,
//
//     out = scale/B * ( U^T V + W^T X ),      U, W: (B, r)   V, X: (B, c),
//
// which is matrix-core work: a workgroup owns a 16x16 tile of `out` and a CHUNK of the batch; its waves split the chunk,
// each walking its part four QPs per v_mfma_*_16x16x4 (operands are read straight from the (B, r) / (B, c) arrays: lane
// (g, c16) holds batch item b0 + g, element 16 I + c16 -- 128 contiguous bytes per 16 lanes, sixteen QPs' operands in
// flight), and the waves' partial tiles are added through LDS IN A FIXED ORDER.
// (round 5) TWO STAGES when the batch is long and the caller provides a workspace: with one workgroup per output tile a
// 10 x 100 gradient at B = 8 192 ran on 7 of 256 CUs in front of the shared-gradient all-reduce; now the batch is cut into
// chunks of whole 256-QP trips, every (tile, chunk) is a workgroup that writes its partial tile to the workspace, and a
// second launch adds the chunks of a tile IN CHUNK ORDER -- the result stays bit-reproducible from run to run, as the
// reference's `.mean(0)` is (no atomics).  Operand loads are unconditional (a lane outside the arrays re-reads the last
// row / column and its value is deselected afterwards): a load whose only use sits under a lane condition is sunk into
// the condition and costs a round trip to memory per element.
#pragma once
#include "qpx_kernels.h"

namespace qpx {

template <class T> struct OuterArgs {
    int B, r, c;
    const T *u, *v, *w, *x;      // v == null: a column of ones (c = 1): the batch mean of u's columns; w == null: no second product
    T scale;          // already divided by B
    T* out;
    T* ws;            // two-stage form: chunks x tiles x 256 partial sums (accumulator layout); else null
    int chunks;       // parts of the batch (1: `out` is written directly)
    int chunk_len;    // QPs per chunk (a multiple of 16 x kOuterWaves)
};
constexpr int kOuterWaves = 16;      // waves per workgroup = parts of a chunk
constexpr int kOuterTrip = 16 * kOuterWaves;      // QPs one trip of a workgroup covers
constexpr int kOuterMaxChunks = 64;

// How a batch is cut (host and tests): chunks of whole trips, enough (tile, chunk) workgroups to fill the chip a few
// times over, never more than kOuterMaxChunks; one chunk = the single-stage form.
QPX_LAYOUT_HD int outer_chunks(int B, int tiles)
{
    const int trips = (B + kOuterTrip - 1) / kOuterTrip;
    int want = (1024 + tiles - 1) / tiles;               // ~four workgroups per CU
    if (want > trips) want = trips;
    if (want > kOuterMaxChunks) want = kOuterMaxChunks;
    return want < 1 ? 1 : want;
}
QPX_LAYOUT_HD int outer_chunk_len(int B, int chunks)
{
    const int trips = (B + kOuterTrip - 1) / kOuterTrip;
    return ((trips + chunks - 1) / chunks) * kOuterTrip;
}

// stage 1 (or the whole job when a.chunks == 1); lds: kOuterWaves x 256 partial tiles
template <class T> QPX_DEV void batch_outer_body(const Block& b, const OuterArgs<T>& a, int tile, int chunk, T* lds)
{
    const int tc = (a.c + 15) >> 4, ntiles = tc * ((a.r + 15) >> 4);
    const int I = tile / tc, J = tile - I * tc;
    const int lane = b.lane(), g = lane >> 4, c16 = lane & 15, wv = b.uniform(b.wave()), nw = b.nwaves();
    const int ri = 16 * I + c16, cj = 16 * J + c16;
    const bool rok = ri < a.r, cok = cj < a.c;
    const int ric = rok ? ri : a.r - 1, cjc = cok ? cj : a.c - 1;         // clamped: every lane loads
    const int b0 = chunk * a.chunk_len, b1 = (b0 + a.chunk_len < a.B) ? b0 + a.chunk_len : a.B;
    // this wave's part of the chunk: multiples of 16 QPs, dealt round-robin (16 QPs = four MFMA pairs per trip)
    T acc[4] = {T(0), T(0), T(0), T(0)};
    const bool hv = a.v != nullptr, hw = a.w != nullptr;      // (uniform: scalar branches around whole loads)
    for (int bb = b0 + 16 * wv; bb < b1; bb += 16 * nw) {
        T au[4], bv[4], aw[4], bx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bi = bb + 4 * k + g;
            const int bic = bi < b1 ? bi : b1 - 1;
            au[k] = a.u[(size_t)bic * a.r + ric];
            bv[k] = hv ? a.v[(size_t)bic * a.c + cjc] : T(1);
            aw[k] = hw ? a.w[(size_t)bic * a.r + ric] : T(0);
            bx[k] = hw ? a.x[(size_t)bic * a.c + cjc] : T(0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool bok = bb + 4 * k + g < b1;
            au[k] = (bok && rok) ? au[k] : T(0);
            aw[k] = (bok && rok) ? aw[k] : T(0);
            bv[k] = (bok && cok) ? bv[k] : T(0);
            bx[k] = (bok && cok) ? bx[k] : T(0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b.mfma16x16x4(au[k], bv[k], acc);
            b.mfma16x16x4(aw[k], bx[k], acc);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) lds[(wv * 4 + rr) * 64 + lane] = acc[rr];
    b.sync();
    if (wv == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            T sum = T(0);
            for (int k = 0; k < nw; ++k) sum += lds[(k * 4 + rr) * 64 + lane];
            if (a.chunks > 1) {
                a.ws[((size_t)chunk * ntiles + tile) * 256 + rr * 64 + lane] = sum;
            } else {
                const int i = 16 * I + Block::mfma_row(T(0), g, rr), j = cj;
                if (i < a.r && j < a.c) a.out[(size_t)i * a.c + j] = a.scale * sum;
            }
        }
    }
}

// stage 2: one workgroup of 256 threads per tile adds the chunks' partial tiles in chunk order
template <class T> QPX_DEV void batch_outer_sum_body(const Block& b, const OuterArgs<T>& a, int tile)
{
    const int tc = (a.c + 15) >> 4, ntiles = tc * ((a.r + 15) >> 4);
    const int I = tile / tc, J = tile - I * tc;
    const int e = b.tid, rr = e >> 6, lane = e & 63, g = lane >> 4, c16 = lane & 15;
    T sum = T(0);
    for (int k = 0; k < a.chunks; ++k) sum += a.ws[((size_t)k * ntiles + tile) * 256 + e];
    const int i = 16 * I + Block::mfma_row(T(0), g, rr), j = 16 * J + c16;
    if (i < a.r && j < a.c) a.out[(size_t)i * a.c + j] = a.scale * sum;
}

// ------------------------------------------------------------------------------------------ small dense solve
// x = M^-1 r for one general k x k system per workgroup (Gaussian elimination with partial pivoting, M and r in global
// memory, both overwritten: r by x).  The one user: factor_solve_kkt_reg with equality constraints (batch.py:273-310), whose
// -eps I in the (y, y) block is a rank-neq correction of the condensed system -- (I + eps Y) dy = dy0 with Y = d(dy)/d(ry)
// -- a neq x neq system per QP that round 4 handed to a library solve on the host side.  Off the QPFunction path and small: one pivot
// per pair of barriers, the trailing update dealt over the 256 threads.
template <class T> struct DenseSolveArgs {
    int B, k;
    T* M;             // (B, k, k), destroyed
    T* r;             // (B, k): right-hand side in, solution out
    int* status;      // QPX_ST_KKT_BREAKDOWN is OR-ed in when a pivot column is exactly zero / not finite (may be null)
};
// lds: 2 k + 257 elements + 257 ints
QPX_LAYOUT_HD size_t dense_solve_lds_elems(int k) { return (size_t)2 * k + 257 + 260; }
template <class T> QPX_DEV void dense_solve_body(const Block& b, const DenseSolveArgs<T>& a, int qp, T* lds)
{
    const int k = a.k, NT = b.nt;
    T* M = a.M + (size_t)qp * k * k;
    T* r = a.r + (size_t)qp * k;
    T* prow = lds;                 // the pivot row, columns j .. k-1
    T* mult = prow + k;            // the multipliers of column j
    T* best = mult + k;            // NT candidates of the pivot search; [256] = the chosen |pivot|
    int* bidx = reinterpret_cast<int*>(best + 257);         // their rows; [256] = the pivot row
    bool ok = true;
    for (int j = 0; j < k; ++j) {
        // pivot search over rows j .. k-1 of column j: thread t scans rows j + t, j + t + NT, ... (first maximum wins)
        T bv = T(-1);
        int bi = j;
        for (int i = j + b.tid; i < k; i += NT) {
            const T v = abs_(M[(size_t)i * k + j]);
            if (v > bv) { bv = v; bi = i; }
        }
        best[b.tid] = bv;
        bidx[b.tid] = bi;
        b.sync();
        if (b.tid == 0) {
            T mv = T(-1);
            int mi = j;
            const int nt = (k - j < NT) ? k - j : NT;
            for (int t = 0; t < nt; ++t)
                if (best[t] > mv || (best[t] == mv && bidx[t] < mi)) { mv = best[t]; mi = bidx[t]; }      // ties: the lowest row
            best[256] = mv;
            bidx[256] = mi;
        }
        b.sync();
        const int pr = bidx[256];
        const T pv_abs = best[256];
        if (!(pv_abs > T(0)) || !finite_(pv_abs)) { ok = false; break; }        // uniform
        // swap rows j and pr (columns j .. k-1 and the right-hand side), publish the pivot row
        for (int c = j + b.tid; c < k; c += NT) {
            const T u = M[(size_t)pr * k + c], l = M[(size_t)j * k + c];
            M[(size_t)j * k + c] = u;
            if (pr != j) M[(size_t)pr * k + c] = l;
            prow[c] = u;
        }
        if (b.tid == 0) {
            const T u = r[pr], l = r[j];
            r[j] = u;
            if (pr != j) r[pr] = l;
        }
        b.sync();
        const T rp = T(1) / prow[j];
        for (int i = j + 1 + b.tid; i < k; i += NT) mult[i] = M[(size_t)i * k + j] * rp;
        b.sync();
        // trailing update: element (i, c), i, c > j; and the right-hand side
        const int w = k - j - 1;
        for (int e = b.tid; e < w * w; e += NT) {
            const int i = j + 1 + e / w, c = j + 1 + e % w;
            M[(size_t)i * k + c] = fma_(-mult[i], prow[c], M[(size_t)i * k + c]);
        }
        const T rj = r[j];
        for (int i = j + 1 + b.tid; i < k; i += NT) r[i] = fma_(-mult[i], rj, r[i]);
        b.sync();
    }
    if (!ok) {
        if (b.tid == 0 && a.status) a.status[qp] |= QPX_ST_KKT_BREAKDOWN;
        for (int i = b.tid; i < k; i += NT) r[i] = Lim<T>::inf() - Lim<T>::inf();
        return;
    }
    // back substitution, one unknown per barrier
    for (int j = k - 1; j >= 0; --j) {
        if (b.tid == 0) r[j] = r[j] / M[(size_t)j * k + j];
        b.sync();
        const T xj = r[j];
        for (int i = b.tid; i < j; i += NT) r[i] = fma_(-M[(size_t)i * k + j], xj, r[i]);
        b.sync();
    }
}

}  // namespace qpx
