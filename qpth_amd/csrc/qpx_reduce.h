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
    const T *u, *v, *w, *x;
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
    for (int bb = b0 + 16 * wv; bb < b1; bb += 16 * nw) {
        T au[4], bv[4], aw[4], bx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bi = bb + 4 * k + g;
            const int bic = bi < b1 ? bi : b1 - 1;
            au[k] = a.u[(size_t)bic * a.r + ric];
            bv[k] = a.v[(size_t)bic * a.c + cjc];
            aw[k] = a.w[(size_t)bic * a.r + ric];
            bx[k] = a.x[(size_t)bic * a.c + cjc];
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

}  // namespace qpx
