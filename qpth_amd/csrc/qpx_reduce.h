// qpx_reduce.h -- batch-mean gradient of a SHARED parameter as one contraction over the batch.
//
// The reference forms B outer products per shared parameter and averages them (qp.py:159-177:
// `dQs = bger(dx, zhat) ...; dQs = dQs.mean(0)`), i.e. it writes and re-reads B x n x n numbers to
// produce n x n.  Summed over the batch the outer products are a dense matrix product with the
// batch as the contraction index,
//
//     out = scale/B * ( U^T V + W^T X ),      U, W: (B, r)   V, X: (B, c),
//
// which is matrix-core work: one wave owns a 16x16 tile of `out` and walks the batch four QPs per
// v_mfma_*_16x16x4 (operands are read straight from the (B, r) / (B, c) arrays: lane (g, c16) holds
// batch item b0 + g, element 16 I + c16 -- 128 contiguous bytes per 16 lanes).  Long batches are split
// over gridDim.y workgroups that add their partial tiles with atomics onto a zeroed `out`.
#pragma once
#include "qpx_kernels.h"

namespace qpx {

template <class T> struct OuterArgs {
    int B, r, c;
    const T *u, *v, *w, *x;
    T scale;          // already divided by B
    T* out;
    int bchunk;       // batch items per workgroup (multiple of 4)
    int use_atomics;  // gridDim.y > 1
};

template <class T> QPX_DEV void batch_outer_body(const Block& b, const OuterArgs<T>& a, int tile, int chunk)
{
    const int tc = (a.c + 15) >> 4;
    const int I = tile / tc, J = tile - I * tc;
    const int lane = b.lane(), g = lane >> 4, c16 = lane & 15;
    const int ri = 16 * I + c16, cj = 16 * J + c16;
    const bool rok = ri < a.r, cok = cj < a.c;
    const int b0 = chunk * a.bchunk;
    const int b1 = (b0 + a.bchunk < a.B) ? b0 + a.bchunk : a.B;
    T acc[4] = {T(0), T(0), T(0), T(0)};
    for (int bb = b0; bb < b1; bb += 4) {
        const int bi = bb + g;
        const bool bok = bi < b1;
        const T au = (bok && rok) ? a.u[(size_t)bi * a.r + ri] : T(0);
        const T bv = (bok && cok) ? a.v[(size_t)bi * a.c + cj] : T(0);
        const T aw = (bok && rok) ? a.w[(size_t)bi * a.r + ri] : T(0);
        const T bx = (bok && cok) ? a.x[(size_t)bi * a.c + cj] : T(0);
        b.mfma16x16x4(au, bv, acc);
        b.mfma16x16x4(aw, bx, acc);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = 16 * I + Block::mfma_row(T(0), g, rr), j = cj;
        if (i < a.r && j < a.c) {
            T* o = a.out + (size_t)i * a.c + j;
            const T val = a.scale * acc[rr];
            if (a.use_atomics) atomic_add_(o, val);
            else *o = val;
        }
    }
}

}  // namespace qpx
