// qpx_reduce.h -- batch-mean gradient of a SHARED parameter as one contraction over the batch.
//
// The reference forms B outer products per shared parameter and averages them (qp.py:159-177:
// `dQs = bger(dx, zhat) ...; dQs = dQs.mean(0)`), i.e. it writes and re-reads B x n x n numbers to
// produce n x n.  Summed over the batch the outer products are a dense matrix product with the
// batch as the contraction index,
//
//     out = scale/B * ( U^T V + W^T X ),      U, W: (B, r)   V, X: (B, c),
//
// which is matrix-core work: a workgroup owns a 16x16 tile of `out`; its waves split the batch, each walking its part
// four QPs per v_mfma_*_16x16x4 (operands are read straight from the (B, r) / (B, c) arrays: lane (g, c16) holds batch
// item b0 + g, element 16 I + c16 -- 128 contiguous bytes per 16 lanes, sixteen QPs' operands in flight), and the
// waves' partial tiles are added through LDS IN A FIXED ORDER: the result is bit-reproducible from run to run, as the
// reference's `.mean(0)` is (round 2 split long batches over workgroups that added with atomics).
#pragma once
#include "qpx_kernels.h"

namespace qpx {

template <class T> struct OuterArgs {
    int B, r, c;
    const T *u, *v, *w, *x;
    T scale;          // already divided by B
    T* out;
};
constexpr int kOuterWaves = 16;      // waves per workgroup = parts of the batch

// lds: kOuterWaves x 256 partial tiles
template <class T> QPX_DEV void batch_outer_body(const Block& b, const OuterArgs<T>& a, int tile, T* lds)
{
    const int tc = (a.c + 15) >> 4;
    const int I = tile / tc, J = tile - I * tc;
    const int lane = b.lane(), g = lane >> 4, c16 = lane & 15, wv = b.uniform(b.wave()), nw = b.nwaves();
    const int ri = 16 * I + c16, cj = 16 * J + c16;
    const bool rok = ri < a.r, cok = cj < a.c;
    // this wave's part of the batch: multiples of 16 QPs, dealt round-robin (16 QPs = four MFMA pairs per trip)
    T acc[4] = {T(0), T(0), T(0), T(0)};
    for (int bb = 16 * wv; bb < a.B; bb += 16 * nw) {
        T au[4], bv[4], aw[4], bx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bi = bb + 4 * k + g;
            const bool bok = bi < a.B;
            au[k] = (bok && rok) ? a.u[(size_t)bi * a.r + ri] : T(0);
            bv[k] = (bok && cok) ? a.v[(size_t)bi * a.c + cj] : T(0);
            aw[k] = (bok && rok) ? a.w[(size_t)bi * a.r + ri] : T(0);
            bx[k] = (bok && cok) ? a.x[(size_t)bi * a.c + cj] : T(0);
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
            const int i = 16 * I + Block::mfma_row(T(0), g, rr), j = cj;
            if (i < a.r && j < a.c) a.out[(size_t)i * a.c + j] = a.scale * sum;
        }
    }
}

}  // namespace qpx
