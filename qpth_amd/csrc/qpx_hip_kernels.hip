// qpx_hip_kernels.hip -- the gfx950 kernels and their launchers.
//
// One workgroup (256 threads = 4 wave64) per QP; the KKT blocks of that QP live in LDS
// (dynamic, up to the full 160 KiB) for the whole kernel, or in the HBM factor blob when they
// do not fit.  Kernels are stream-ordered, allocate nothing and never synchronise the host.
//
// Compiled once per (QPX_TU_KERNEL, QPX_TU_REAL): 5 = sweep pre-factorisation (16x16 thread grid),
// (1, 2, 3 were the round-1 workgroup kernels, deleted in round 5)
// 6 = ipm (thread grid), 7 = kkt/backward (thread grid), 8 = ipm (8x8 thread grid = one wave), 9 = ipm (matrix-core tiles, f64 only),
// 10 = batch-mean outer products of shared-parameter gradients, 11 = the large-QP family (qpx_big.h),
// 13 = the finishing stage on matrix-core tiles (f64 only; its thread-grid form lives in 7), 14 = pre_factor_kkt on matrix-core tiles (f64 only).
// (15 was qpx_forward as ONE launch, round 5 -- 14's body and 9's chain-wave loop in one workgroup: parity-green, 7 % SLOWER at C2
// and 39 % at B = 8192 n = m = 64, deleted: profiles/r05a_ab_forward_one_launch_*.txt.)  (12 was the sweep
// pre-factorisation on matrix-core tiles of round 3: parity-green, never faster than the thread-grid sweep, deleted in round 4.)
#include <hip/hip_runtime.h>

#include "../../include/qpx.h"
#include "qpx_launch.h"

#ifndef QPX_TU_KERNEL
#error "QPX_TU_KERNEL and QPX_TU_REAL (float|double) must be defined"
#endif

namespace qpx {

// Dynamic LDS above 64 KiB has to be opted into once per kernel symbol AND DEVICE (the attribute belongs to the
// device's copy of the code object): one flag per device of the process.
constexpr int kMaxDevFlags = 16;
struct BigLdsFlags { bool done[kMaxDevFlags] = {}; };
template <class K> static int allow_big_lds(K kernel, size_t bytes, BigLdsFlags& flags)
{
    if (bytes <= 64 * 1024) return QPX_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return QPX_ERR_LAUNCH;
    const bool tracked = dev < kMaxDevFlags;          // devices beyond the table opt in on every launch (the call is idempotent)
    if (tracked && flags.done[dev]) return QPX_OK;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kMaxLdsBytes) != hipSuccess)
        return QPX_ERR_LAUNCH;
    if (tracked) flags.done[dev] = true;   // only after success; idempotent, so a benign race between host threads is harmless
    return QPX_OK;
}

#if QPX_TU_KERNEL == 5
// (at least two workgroups per CU -- <= 256 registers -- at every size: the largest instantiation holds 91 matrix entries per thread)
template <class T, int NBL> __global__ __launch_bounds__(256, 2) void k_sweep(PrefactorArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    sweep_body<T, NBL>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T, int NBL> int launch_sweep(const PrefactorArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_sweep<T, NBL>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#define QPX_INSTG(NBL) template int launch_sweep<QPX_TU_REAL, NBL>(const PrefactorArgs<QPX_TU_REAL>&, size_t, void*);
QPX_INSTG(1) QPX_INSTG(2) QPX_INSTG(4) QPX_INSTG(7) QPX_INSTG(8) QPX_INSTG(10) QPX_INSTG(13)
#elif QPX_TU_KERNEL == 6
// two workgroups per CU (2 waves per SIMD) for the common sizes: the two QPs hide each other's
// barrier / LDS latencies
template <class T, int NBL, int NS> __global__ __launch_bounds__(256, (NBL <= 7 ? 2 : 1)) void k_ipm_grid(IpmArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    ipm_grid_body<T, 16, NBL, NS>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T, int NBL, int NS> int launch_ipm_grid(const IpmArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_ipm_grid<T, NBL, NS>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#define QPX_INSTG(NBL, NS) template int launch_ipm_grid<QPX_TU_REAL, NBL, NS>(const IpmArgs<QPX_TU_REAL>&, size_t, void*);
QPX_INSTG(1, 1) QPX_INSTG(1, 2) QPX_INSTG(1, 4) QPX_INSTG(2, 1) QPX_INSTG(2, 2) QPX_INSTG(2, 4)
QPX_INSTG(4, 1) QPX_INSTG(4, 2) QPX_INSTG(4, 4) QPX_INSTG(7, 2) QPX_INSTG(7, 4) QPX_INSTG(10, 4) QPX_INSTG(13, 4)
#elif QPX_TU_KERNEL == 7
template <class T, int NBL, bool kBw> __global__ __launch_bounds__(256, (NBL <= 7 ? 2 : 1)) void k_kkt_grid(KktArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    kkt_grid_body<T, 16, NBL, kBw>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T, int NBL, bool kBw> int launch_kkt_grid(const KktArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_kkt_grid<T, NBL, kBw>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#define QPX_INSTG(NBL)                                                                              \
    template int launch_kkt_grid<QPX_TU_REAL, NBL, false>(const KktArgs<QPX_TU_REAL>&, size_t, void*); \
    template int launch_kkt_grid<QPX_TU_REAL, NBL, true>(const KktArgs<QPX_TU_REAL>&, size_t, void*);
QPX_INSTG(1) QPX_INSTG(2) QPX_INSTG(4) QPX_INSTG(7) QPX_INSTG(10) QPX_INSTG(13)
// the finishing stage (qpx_polish) on the thread grid
template <class T, int NBL> __global__ __launch_bounds__(256) void k_polish_grid(PolishArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    polish_grid_body<T, 16, NBL>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T, int NBL> int launch_polish_grid(const PolishArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_polish_grid<T, NBL>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#define QPX_INSTP(NBL) template int launch_polish_grid<QPX_TU_REAL, NBL>(const PolishArgs<QPX_TU_REAL>&, size_t, void*);
QPX_INSTP(1) QPX_INSTP(2) QPX_INSTP(4) QPX_INSTP(7) QPX_INSTP(10) QPX_INSTP(13)
#elif QPX_TU_KERNEL == 8
template <class T, int NBL, int NS> __global__ __launch_bounds__(64) void k_ipm_grid8(IpmArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    ipm_grid_body<T, 8, NBL, NS>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T, int NBL, int NS> int launch_ipm_grid8(const IpmArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_ipm_grid8<T, NBL, NS>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(64), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#define QPX_INSTG(NBL, NS) template int launch_ipm_grid8<QPX_TU_REAL, NBL, NS>(const IpmArgs<QPX_TU_REAL>&, size_t, void*);
QPX_INSTG(2, 1) QPX_INSTG(2, 2) QPX_INSTG(4, 1) QPX_INSTG(4, 2) QPX_INSTG(8, 1) QPX_INSTG(8, 2) QPX_INSTG(13, 2)
#elif QPX_TU_KERNEL == 13
// the finishing stage (qpx_polish) on matrix-core tiles, f64: the forms the dispatcher picks by default
template <int NBL, int NW, bool CH> __global__ __launch_bounds__(64 * NW, 2) void k_polish_tile(PolishArgs<double> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    polish_mat_body<double, TileMat<NBL, NW, CH>>(b, a, (int)blockIdx.x, reinterpret_cast<double*>(qpx_smem));
}
template <int NBL, int NW, bool CH> int launch_polish_tile(const PolishArgs<double>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_polish_tile<NBL, NW, CH>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(64 * NW), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#define QPX_INSTPT(NBL, NW, CH) template int launch_polish_tile<NBL, NW, CH>(const PolishArgs<double>&, size_t, void*);
QPX_INSTPT(1, 1, false) QPX_INSTPT(2, 1, false) QPX_INSTPT(4, 1, false) QPX_INSTPT(4, 4, true) QPX_INSTPT(7, 4, true)
#elif QPX_TU_KERNEL == 14
// pre_factor_kkt on matrix-core tiles (qpx_prefac.h), f64, neq = 0: four waves per QP, two QPs per CU
template <int NBN, bool kEq> __global__ __launch_bounds__(256, 2) void k_prefac_tile(PrefactorArgs<double> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    prefac_tile_body<NBN, kEq>(b, a, (int)blockIdx.x, reinterpret_cast<double*>(qpx_smem));
}
template <int NBN, bool kEq> int launch_prefac_tile(const PrefactorArgs<double>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_prefac_tile<NBN, kEq>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(256), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
template int launch_prefac_tile<4, false>(const PrefactorArgs<double>&, size_t, void*);
template int launch_prefac_tile<7, false>(const PrefactorArgs<double>&, size_t, void*);
template int launch_prefac_tile<4, true>(const PrefactorArgs<double>&, size_t, void*);
template int launch_prefac_tile<7, true>(const PrefactorArgs<double>&, size_t, void*);
#elif QPX_TU_KERNEL == 10 || QPX_TU_KERNEL == 11
// defined below, outside the launcher chain
#elif QPX_TU_KERNEL == 9
// NW waves per QP, always at least two waves per SIMD (<= 256 registers per lane): at that occupancy the
// compiler keeps MFMA accumulators in VGPRs.  (At one wave per SIMD it moves every tile through AGPRs --
// 16 copies and a full-latency stall per MFMA -- and the flag that forbids it, -amdgpu-mfma-vgpr-form,
// crashes clang 22 on some instantiations; so the one-wave form is not built for NBL = 7, whose 28
// tiles alone are 224 registers.)
template <int NBL, int NW, int NS, bool CH>
__global__ __launch_bounds__(64 * NW, 2) void k_ipm_tile(IpmArgs<double> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    ipm_tile_body<NBL, NW, NS, CH>(b, a, (int)blockIdx.x, reinterpret_cast<double*>(qpx_smem));
}
template <int NBL, int NW, int NS, bool CH> int launch_ipm_tile(const IpmArgs<double>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_ipm_tile<NBL, NW, NS, CH>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(64 * NW), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#ifdef QPX_PANEL_PROF
// profiling build only: read and reset the panel sub-phase counters (see qpx_tile.h)
extern "C" int qpx_panel_prof_read(unsigned long long* out)
{
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(qpx_panel_prof), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(qpx_panel_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
extern "C" int qpx_chain_prof_read(unsigned long long* out)      // 20 counters of the chain-wave form
{
    unsigned long long zero[20] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(qpx_chain_prof), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(qpx_chain_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif
template <int NBL, int NW, bool kBw, bool CH>
__global__ __launch_bounds__(64 * NW, 2) void k_kkt_tile(KktArgs<double> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    kkt_tile_body<NBL, NW, kBw, CH>(b, a, (int)blockIdx.x, reinterpret_cast<double*>(qpx_smem));
}
template <int NBL, int NW, bool kBw, bool CH> int launch_kkt_tile(const KktArgs<double>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_kkt_tile<NBL, NW, kBw, CH>;
    static BigLdsFlags big_lds_enabled;
    if (allow_big_lds(kern, lds_bytes, big_lds_enabled)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(64 * NW), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
#define QPX_INSTK(NBL, NW, CH)                                                                     \
    template int launch_kkt_tile<NBL, NW, false, CH>(const KktArgs<double>&, size_t, void*);       \
    template int launch_kkt_tile<NBL, NW, true, CH>(const KktArgs<double>&, size_t, void*);
#if !defined(QPX_TILE_ONLY)
QPX_INSTK(1, 1, false) QPX_INSTK(2, 1, false) QPX_INSTK(4, 1, false) QPX_INSTK(4, 2, false) QPX_INSTK(7, 2, false)
QPX_INSTK(7, 4, true) QPX_INSTK(4, 4, true)
#endif
#define QPX_INSTT(NBL, NW, NS, CH) template int launch_ipm_tile<NBL, NW, NS, CH>(const IpmArgs<double>&, size_t, void*);
#if defined(QPX_TILE_ONLY)
QPX_INSTT(7, QPX_TILE_ONLY, 2, QPX_TILE_ONLY == 4)
#else
QPX_INSTT(1, 1, 1, false) QPX_INSTT(1, 1, 2, false) QPX_INSTT(1, 1, 4, false) QPX_INSTT(2, 1, 1, false) QPX_INSTT(2, 1, 2, false)
QPX_INSTT(2, 1, 4, false) QPX_INSTT(4, 1, 1, false) QPX_INSTT(4, 1, 2, false) QPX_INSTT(4, 1, 4, false) QPX_INSTT(4, 2, 1, false)
QPX_INSTT(4, 2, 2, false) QPX_INSTT(4, 2, 4, false) QPX_INSTT(7, 2, 2, false) QPX_INSTT(7, 2, 4, false)
QPX_INSTT(7, 4, 2, true) QPX_INSTT(7, 4, 4, true) QPX_INSTT(4, 4, 1, true) QPX_INSTT(4, 4, 2, true) QPX_INSTT(4, 4, 4, true)
#endif
#endif

#if QPX_TU_KERNEL == 11
// ---- the large-QP family (qpx_big.h): grid (B, chunks), 256 threads (phase kernel: one wave)
#define QPX_BIG_KERNEL(NAME, ARGS, BODY, THREADS)                                                        \
    template <class T> __global__ __launch_bounds__(THREADS) void NAME(ARGS<T> a)                        \
    {                                                                                                    \
        extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];                         \
        const Block b{(int)threadIdx.x, (int)blockDim.x};                                                \
        BODY;                                                                                            \
    }
QPX_BIG_KERNEL(k_big_pack, BigPackArgs, (big_pack_body<T>(b, a, (int)blockIdx.x, (int)blockIdx.y)), 256)
QPX_BIG_KERNEL(k_big_panel, BigPanelArgs, (big_panel_body<T>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem))), 256)
// GEMM tiles, XCD-aware: workgroup id -> XCD id % 8 (observed dispatch rule, a speed assumption only).  A QP's tiles
// re-read the same operand panels, so they should run at the same time on ONE XCD (one L2): with B a multiple of 8
// the grid is 1-D, XCD x works through the QPs x, x + 8, ... one after the other, tile index fastest.  (The plain
// (qp, tile) grid puts a QP on one XCD too but runs tile t of sixteen QPs side by side: sixteen panel sets per L2.)
template <class T> QPX_DEV void big_gemm_where(const BigGemmArgs<T>& a, int ntiles, int swz, int& qp, int& tile)
{
    qp = (int)blockIdx.x; tile = (int)blockIdx.y;
    if (swz) {
        int id = (int)blockIdx.x;
        if (a.fuse) {
            // tile 0 goes on to eliminate a diagonal block (the longest job of the launch): all of them first
            if (id < a.B) { tile = 0; }
            else {
                id -= a.B;
                const int slot = id >> 3;
                qp = (id & 7) + 8 * (slot / (ntiles - 1));
                tile = 1 + slot % (ntiles - 1);
            }
        } else {
            const int slot = id >> 3;
            qp = (id & 7) + 8 * (slot / ntiles);
            tile = slot % ntiles;
        }
    }
}
// the tile product, pipelined (round 4): 37 KB of LDS and <= 128 registers, four workgroups per CU (a launch that also
// eliminates a diagonal block: 71 KB, two)
template <class T, bool kFuse> __global__ __launch_bounds__(256, (kFuse ? 2 : 4)) void k_big_gemm2(BigGemmArgs<T> a, int ntiles, int swz)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    int qp, tile;
    big_gemm_where(a, ntiles, swz, qp, tile);
    big_gemm2_body<T, kFuse>(b, a, qp, tile, reinterpret_cast<T*>(qpx_smem));
}
template <class T, bool kLong = false> __global__ __launch_bounds__(64 * kTrsvNW) void k_big_trsv(BigTrsvArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    big_trsv_body<T, kLong>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
QPX_BIG_KERNEL(k_big_gemv, BigGemvArgs, (big_gemv_body<T>(b, a, (int)blockIdx.x, (int)blockIdx.y, reinterpret_cast<T*>(qpx_smem))), 256)
QPX_BIG_KERNEL(k_big_symv, BigSymvArgs, (big_symv_body<T>(b, a, (int)blockIdx.x, (int)blockIdx.y, reinterpret_cast<T*>(qpx_smem))), 256)
QPX_BIG_KERNEL(k_big_vec, BigVecArgs, (big_vec_body<T>(b, a, (int)blockIdx.x)), 256)
QPX_BIG_KERNEL(k_big_kkt, BigKktArgs, (big_kkt_body<T>(b, a, (int)blockIdx.x, (int)blockIdx.y)), 256)
template <class T, int NS> __global__ __launch_bounds__(64) void k_big_phase(BigPhaseArgs<T> a)
{
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    big_phase_body<T, NS>(b, a, (int)blockIdx.x);
}
template <class T, int NS> __global__ __launch_bounds__(64 * kTrsvNW) void k_big_solve(BigSolveArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    big_solve_body<T, NS>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T, int NS> __global__ __launch_bounds__(256) void k_big_diag(BigDiagArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    big_diag_body<T, NS>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T> __global__ __launch_bounds__(64 * kBigPolWaves) void k_big_polish(BigPolishArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    big_polish_body<T>(b, a, (int)blockIdx.x, reinterpret_cast<double*>(qpx_smem));
}
template <class K, class A> static int big_launch(K kern, const A& a, int gx, int gy, int threads, size_t lds, void* stream, BigLdsFlags& big_ok)
{
    if (allow_big_lds(kern, lds, big_ok)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(threads), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
template <class T> int launch_big_pack(const BigPackArgs<T>& a, int gy, void* s) { static BigLdsFlags f; return big_launch(k_big_pack<T>, a, a.B, gy, 256, 0, s, f); }
template <class T> int launch_big_panel(const BigPanelArgs<T>& a, void* s) { static BigLdsFlags f; return big_launch(k_big_panel<T>, a, a.B, 1, 256, big_panel_lds_elems() * sizeof(T), s, f); }
template <class T> int launch_big_gemm(const BigGemmArgs<T>& a, void* s)
{
    static BigLdsFlags f2;
    const int ntiles = a.nti * a.ntj, swz = (a.B % 8 == 0 && ntiles > 1 && a.fuse && !a.no_swizzle) ? 1 : 0;   // measured (r02i): the trailing updates gain 3 %, R = Zt Zt^T (half its tiles empty) loses 30 %
    const size_t lds = big_gemm2_lds_elems<T>(a.fuse != 0, a.mirror != 0) * sizeof(T);
    if (a.fuse) {
        if (allow_big_lds(k_big_gemm2<T, true>, lds, f2)) return QPX_ERR_LAUNCH;
        hipLaunchKernelGGL((k_big_gemm2<T, true>), swz ? dim3(a.B * ntiles) : dim3(a.B, ntiles), dim3(256), lds, (hipStream_t)s, a, ntiles, swz);
    } else {
        hipLaunchKernelGGL((k_big_gemm2<T, false>), swz ? dim3(a.B * ntiles) : dim3(a.B, ntiles), dim3(256), lds, (hipStream_t)s, a, ntiles, swz);
    }
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
template <class T> int launch_big_trsv(const BigTrsvArgs<T>& a, void* s)
{
    static BigLdsFlags f;
    if (a.nb > 8) { static BigLdsFlags f2; return big_launch(k_big_trsv<T, true>, a, a.B, 1, 64 * kTrsvNW, big_trsv_lds_elems(a.nb * kBB) * sizeof(T), s, f2); }
    return big_launch(k_big_trsv<T, false>, a, a.B, 1, 64 * kTrsvNW, big_trsv_lds_elems(a.nb * kBB) * sizeof(T), s, f);
}
template <class T> int launch_big_gemv(const BigGemvArgs<T>& a, void* s)
{
    static BigLdsFlags f;
    const int outs = a.trans ? a.cols : a.rows;
    return big_launch(k_big_gemv<T>, a, a.B, (outs + kBB - 1) / kBB, 256, big_gemv_lds_elems(a.trans ? a.rows : a.cols) * sizeof(T), s, f);
}
template <class T> int launch_big_symv(const BigSymvArgs<T>& a, void* s)
{
    static BigLdsFlags f;
    return big_launch(k_big_symv<T>, a, a.B, a.stage == 0 ? a.ld / kBB : 1, 256, big_symv_lds_elems() * sizeof(T), s, f);
}
template <class T> int launch_big_vec(const BigVecArgs<T>& a, void* s) { static BigLdsFlags f; return big_launch(k_big_vec<T>, a, a.B, 1, 256, 0, s, f); }
template <class T> int launch_big_kkt(const BigKktArgs<T>& a, int gy, void* s) { static BigLdsFlags f; return big_launch(k_big_kkt<T>, a, a.B, gy, 256, 0, s, f); }
template <class T> int launch_big_phase(const BigPhaseArgs<T>& a, void* s)
{
    static BigLdsFlags f;
    const int ns = big_pad(a.m) / kWave;
    switch (ns) {
    case 1: return big_launch(k_big_phase<T, 1>, a, a.B, 1, 64, 0, s, f);
    case 2: return big_launch(k_big_phase<T, 2>, a, a.B, 1, 64, 0, s, f);
    case 3: case 4: return big_launch(k_big_phase<T, 4>, a, a.B, 1, 64, 0, s, f);
    case 5: case 6: case 7: case 8: return big_launch(k_big_phase<T, 8>, a, a.B, 1, 64, 0, s, f);
    default: return big_launch(k_big_phase<T, 16>, a, a.B, 1, 64, 0, s, f);
    }
}
template <class T> int launch_big_solve(const BigSolveArgs<T>& a, void* s)
{
    static BigLdsFlags f;
    const size_t lds = big_trsv_lds_elems(a.t.nb * kBB) * sizeof(T);
    const int ns = big_pad(a.ph.m) / kWave;
    switch (ns) {
    case 1: return big_launch(k_big_solve<T, 1>, a, a.t.B, 1, 64 * kTrsvNW, lds, s, f);
    case 2: return big_launch(k_big_solve<T, 2>, a, a.t.B, 1, 64 * kTrsvNW, lds, s, f);
    case 3: case 4: return big_launch(k_big_solve<T, 4>, a, a.t.B, 1, 64 * kTrsvNW, lds, s, f);
    case 5: case 6: case 7: case 8: return big_launch(k_big_solve<T, 8>, a, a.t.B, 1, 64 * kTrsvNW, lds, s, f);
    default: return big_launch(k_big_solve<T, 16>, a, a.t.B, 1, 64 * kTrsvNW, lds, s, f);
    }
}
template <class T> int launch_big_diag(const BigDiagArgs<T>& a, void* s)
{
    static BigLdsFlags f;
    const size_t lds = big_panel_lds_elems() * sizeof(T);
    const int ns = big_pad(a.ph.m) / kWave;
    switch (ns) {
    case 1: return big_launch(k_big_diag<T, 1>, a, a.p.B, 1, 256, lds, s, f);
    case 2: return big_launch(k_big_diag<T, 2>, a, a.p.B, 1, 256, lds, s, f);
    case 3: case 4: return big_launch(k_big_diag<T, 4>, a, a.p.B, 1, 256, lds, s, f);
    case 5: case 6: case 7: case 8: return big_launch(k_big_diag<T, 8>, a, a.p.B, 1, 256, lds, s, f);
    default: return big_launch(k_big_diag<T, 16>, a, a.p.B, 1, 256, lds, s, f);
    }
}
template <class T> int launch_big_polish(const BigPolishArgs<T>& a, void* s)
{
    static BigLdsFlags f;
    const BigLayout L = big_layout(a.n, a.m, a.q);
    return big_launch(k_big_polish<T>, a, a.B, 1, 64 * kBigPolWaves, big_polish_lds_doubles(L.VP) * sizeof(double), s, f);
}
#define QPX_INSTB(NAME, ARGS) template int NAME<QPX_TU_REAL>(const ARGS<QPX_TU_REAL>&, void*);
QPX_INSTB(launch_big_polish, BigPolishArgs)
QPX_INSTB(launch_big_solve, BigSolveArgs) QPX_INSTB(launch_big_diag, BigDiagArgs)
template int launch_big_pack<QPX_TU_REAL>(const BigPackArgs<QPX_TU_REAL>&, int, void*);
template int launch_big_kkt<QPX_TU_REAL>(const BigKktArgs<QPX_TU_REAL>&, int, void*);
QPX_INSTB(launch_big_panel, BigPanelArgs) QPX_INSTB(launch_big_gemm, BigGemmArgs) QPX_INSTB(launch_big_trsv, BigTrsvArgs) QPX_INSTB(launch_big_gemv, BigGemvArgs)
QPX_INSTB(launch_big_vec, BigVecArgs) QPX_INSTB(launch_big_phase, BigPhaseArgs) QPX_INSTB(launch_big_symv, BigSymvArgs)
#endif

#if QPX_TU_KERNEL == 10
template <class T> __global__ __launch_bounds__(64 * kOuterWaves) void k_batch_outer(OuterArgs<T> a)
{
    __shared__ T lds[kOuterWaves * 256];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    batch_outer_body<T>(b, a, (int)blockIdx.x, (int)blockIdx.y, lds);
}
template <class T> __global__ __launch_bounds__(256) void k_batch_outer_sum(OuterArgs<T> a)
{
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    batch_outer_sum_body<T>(b, a, (int)blockIdx.x);
}
template <class T> int launch_batch_outer(const OuterArgs<T>& a, int tiles, void* stream)
{
    hipLaunchKernelGGL(k_batch_outer<T>, dim3(tiles, a.chunks), dim3(64 * kOuterWaves), 0, (hipStream_t)stream, a);
    if (a.chunks > 1) hipLaunchKernelGGL(k_batch_outer_sum<T>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
template int launch_batch_outer<QPX_TU_REAL>(const OuterArgs<QPX_TU_REAL>&, int, void*);
template <class T> __global__ __launch_bounds__(256) void k_dense_solve(DenseSolveArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    dense_solve_body<T>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}
template <class T> int launch_dense_solve(const DenseSolveArgs<T>& a, void* stream)
{
    hipLaunchKernelGGL(k_dense_solve<T>, dim3(a.B), dim3(256), dense_solve_lds_elems(a.k) * sizeof(T), (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}
template int launch_dense_solve<QPX_TU_REAL>(const DenseSolveArgs<QPX_TU_REAL>&, void*);
#endif

}  // namespace qpx
