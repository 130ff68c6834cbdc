// qpx_layout.h -- HBM layout of the per-QP "factor blob" and shared constants.
//
// The blob is what the reference keeps in (Q_LU, S_LU, R) between pre_factor_kkt, the IPM loop
// and backward (qpth/solvers/pdipm/batch.py:375-429, qpth/qp.py:93,150-155).  One blob per QP,
// `fac_stride` elements apart, all sub-arrays 16-byte aligned.  There are two families, chosen by
// the dispatcher from (dtype, n, m, q) alone (qpx_api.inc: blob_images):
//
// (a) thread-grid / matrix-core kernels, n+q+m <= 208 (`images` != 0): what the symmetric sweep of
//     the augmented matrix leaves behind (qpx_grid.h: sweep_body) -- no triangular factors at all:
//
//   Kneg   n x n      -K,  K = Q^-1 - Q^-1 A^T S11^-1 A Q^-1  (both triangles)
//   MT     n x m      (G K)^T: M p is a column-parallel sum over its rows, M^T z a row dot
//   NTn    q x n      -N^T,  N = Q^-1 A^T S11^-1
//   W      m x q      G N
//   S11i   q x q      (A Q^-1 A^T)^-1
//   scal   4          [0] = || G^T 1 ||_2 ; [4..11] phase timers of the profiling build
//   R = G K G^T (the reference's R, batch.py:396-399,424) in the REGISTER IMAGE(S) its consumers load,
//   `images` bit 1: Rg  16x16 thread grid, entry [(li(li+1)/2 + lj)*256 + a + 16b] = R[16li+a][16lj+b]
//            bit 2: Rw  8x8 thread grid (one wave), entry [(li(li+1)/2 + lj)*64 + a + 8b]
//            bit 4: Rm  16x16 tiles in the C/D layout of v_mfma_f64_16x16x4 (tile_image_index)
//   (f64 with m <= 112: Rm only; otherwise Rg + Rw).
//
// (b) the large-QP family (qpx_big.h), every larger size up to 512 per dimension (`images` == 8): BigLayout there --
//     Cholesky factors and the projected Zt in 64 x 64 blocks, the loop's vectors, the finishing stage's iterates.
//
// (Until round 5 a third family held the packed Cholesky factors of the round-1 workgroup kernels, `images` == 0.)
#pragma once
#include <cstddef>

namespace qpx {

#if defined(__HIPCC__)
#define QPX_LAYOUT_HD __host__ __device__ inline
#else
#define QPX_LAYOUT_HD inline
#endif

QPX_LAYOUT_HD size_t tri(size_t i) { return i * (i + 1) / 2; }
QPX_LAYOUT_HD int tri(int i) { return i * (i + 1) / 2; }   // 32-bit form for kernel index math
QPX_LAYOUT_HD size_t align4(size_t x) { return (x + 3) & ~(size_t)3; }

// Number of 8-row blocks the 8x8 thread-grid (one wave per QP) loop kernel is instantiated with for
// nineq = m (0: not available).
QPX_LAYOUT_HD int wave_nb(int m)
{
    const int need = (m + 7) / 8;
    if (need <= 2) return 2;
    if (need <= 4) return 4;
    if (need <= 8) return 8;
    if (need <= 13) return 13;
    return 0;
}

// Blocks of 16 rows the 16x16 thread-grid kernels (qpx_grid.h) use for a matrix of order `ord`
// (0: not instantiated for this size).
QPX_LAYOUT_HD int grid_nb(int ord)
{
    const int need = (ord + 15) / 16;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 4) return 4;
    if (need <= 7) return 7;
    if (need <= 10) return 10;
    if (need <= 13) return 13;
    return 0;
}

// ---- matrix-core tile kernels (qpx_tile.h): R as 16x16 tiles of the lower block triangle, each
// in the register layout of the f64 MFMA accumulator: element (i, j), tile (I, J) = (i/16, j/16),
// J <= I, is entry (I (I+1)/2 + J) * 256 + ((i%16) / 4) * 64 + ((i%16) % 4) * 16 + j%16.
QPX_LAYOUT_HD int tile_nb(int m)      // tile rows the kernels are instantiated with for order m (0: n/a)
{
    const int need = (m + 15) / 16;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 4) return 4;
    if (need <= 7) return 7;
    return 0;
}
QPX_LAYOUT_HD size_t tile_image_elems(int nbl) { return (size_t)(nbl * (nbl + 1) / 2) * 256; }
QPX_LAYOUT_HD size_t tile_image_index(int i, int j)   // i >= j, or both in the same diagonal tile
{
    const int I = i >> 4, J = j >> 4, ri = i & 15, c = j & 15;
    return ((size_t)(I * (I + 1) / 2 + J) * 4 + (ri >> 2)) * 64 + (size_t)((ri & 3) * 16 + c);
}

// Blocks of 16 the sweep pre-factorisation is instantiated with for the augmented order n+q+m: the
// grid_nb list plus 8 (C5: 64 + 64 = 128 is exactly 8 blocks; rounding it up to 10 would sweep 55 blocks
// per thread instead of 36).
QPX_LAYOUT_HD int sweep_nb(int ord)
{
    const int need = (ord + 15) / 16;
    return need == 8 ? 8 : grid_nb(ord);
}

// (The fields and the `images == 0` branch of the packed-Cholesky family are DEAD since round 5 -- blob_images never returns 0
// -- and stay on purpose: fac_layout is evaluated inside every kernel with a run-time `images`, and without the branch the
// compiler allocates the registers of the C2 loop kernel differently: same source otherwise, 15 450 instead of 15 788
// instructions, and 0.473 instead of 0.468 ms on the same box (profiles/r05z_ab_r04.txt).  Tidiness is not worth 1 %.)
struct FacLayout {
    // family (b): dead, see above
    size_t L, dinvL, Zp, R, Yh, V, L11, dinv11, r1, T;
    // family (a)
    size_t Kneg, MT, NTn, W, S11i, Rg, Rw, Rm;
    size_t scal, prof;
    size_t total;
    int images;   // 0: family (b); else bit mask of the R images present (1 Rg, 2 Rw, 4 Rm)
    int nbw;      // 8x8-grid blocks of 8 for m (0 = n/a)
    int nbg;      // grid blocks of 16 for m
    int nba;      // grid blocks of 16 for the augmented order n+q+m (0 = family (a) unavailable)
    int nbt;      // tile rows of 16 for m in the matrix-core kernels (0 = n/a)
};

// can the thread-grid / tile kernels run this size at all?
QPX_LAYOUT_HD bool grid_family_fits(int n, int m, int q) { return grid_nb(n + q + m) > 0; }

QPX_LAYOUT_HD FacLayout fac_layout(int n, int m, int q, int images)
{
    FacLayout f;
    size_t o = 0;
    f.images = images;
    f.L = f.dinvL = f.Zp = f.R = f.Yh = f.V = f.L11 = f.dinv11 = f.r1 = f.T = 0;
    f.Kneg = f.MT = f.NTn = f.W = f.S11i = f.Rg = f.Rw = f.Rm = 0;
    f.nbw = f.nbg = f.nba = f.nbt = 0;
    if (images == 0) {
        f.L = o;      o += align4(tri((size_t)n));
        f.dinvL = o;  o += align4(n);
        f.Zp = o;     o += align4((size_t)n * m);
        f.R = o;      o += align4(tri((size_t)m));
        f.Yh = o;     o += align4((size_t)n * q);
        f.V = o;      o += align4((size_t)q * m);
        f.L11 = o;    o += align4(tri((size_t)q));
        f.dinv11 = o; o += align4(q);
        f.r1 = o;     o += align4(m);
        f.scal = o;   o += 4;
        f.prof = o;   o += 8;
        f.T = o;      o += align4(tri((size_t)m));
    } else {
        f.nba = grid_nb(n + q + m);
        f.nbg = grid_nb(m);
        f.nbw = wave_nb(m);
        f.nbt = tile_nb(m);
        f.Kneg = o; o += align4((size_t)n * n);
        f.MT = o;   o += align4((size_t)n * m);
        f.NTn = o;  o += align4((size_t)q * n);
        f.W = o;    o += align4((size_t)m * q);
        f.S11i = o; o += align4((size_t)q * q);
        f.scal = o; o += 4;
        f.prof = o; o += 8;
        f.Rg = o;   if (images & 1) o += (size_t)(f.nbg * (f.nbg + 1) / 2) * 256;
        f.Rw = o;   if ((images & 2) && f.nbw > 0) o += (size_t)(f.nbw * (f.nbw + 1) / 2) * 64;
        f.Rm = o;   if ((images & 4) && f.nbt > 0) o += tile_image_elems(f.nbt);
    }
    f.total = o;
    return f;
}

// per-QP status bits written by the kernels (device memory, int32)
enum : int {
    QPX_ST_Q_NOT_SPD = 1,     // Cholesky of Q broke down      -> RuntimeError('Q is not SPD.')
    QPX_ST_A_RANK = 2,        // Cholesky of A Q^-1 A^T broke down (A row-rank deficient)
    QPX_ST_KKT_BREAKDOWN = 4, // factor_kkt broke down during the IPM loop (best iterate returned)
    QPX_ST_INACCURATE = 8,    // best residual > 1                 -> INACC_ERR warning
    QPX_ST_MAXITER = 16,      // loop ended on maxIter
    QPX_ST_NONFINITE = 32     // iterate went NaN/Inf (best iterate returned)
};

}  // namespace qpx
