// qpx_hip.hip -- gfx950 kernels + launchers behind the C ABI of include/qpx.h.
//
// One workgroup (256 threads = 4 wave64) per QP; the KKT blocks of that QP live in LDS
// (dynamic, up to the full 160 KiB) for the whole kernel, or in the HBM factor blob when they
// do not fit.  Kernels are stream-ordered, allocate nothing and never synchronise the host.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/qpx.h"
#include "qpx_kernels.h"

namespace qpx {

template <int V> using Int = std::integral_constant<int, V>;
template <bool V> using Bool = std::integral_constant<bool, V>;

constexpr int kThreads = 256;

inline size_t lds_budget_bytes() { return kMaxLdsBytes; }

template <class T, int NS, bool kLds>
__global__ __launch_bounds__(kThreads) void k_prefactor(PrefactorArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    prefactor_body<T, NS, kLds>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}

template <class T, int NS, bool kLds>
__global__ __launch_bounds__(kThreads) void k_ipm(IpmArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    ipm_body<T, NS, kLds>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}

template <class T, int NS, bool kLds, bool kBw>
__global__ __launch_bounds__(kThreads) void k_kkt(KktArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qpx_smem[];
    const Block b{(int)threadIdx.x, (int)blockDim.x};
    kkt_body<T, NS, kLds, kBw>(b, a, (int)blockIdx.x, reinterpret_cast<T*>(qpx_smem));
}

// Dynamic LDS above 64 KiB has to be opted into once per kernel symbol.
template <class K> static int allow_big_lds(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return QPX_OK;
    static thread_local const void* last = nullptr;
    if (last == (const void*)kernel) return QPX_OK;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kMaxLdsBytes) != hipSuccess)
        return QPX_ERR_LAUNCH;
    last = (const void*)kernel;
    return QPX_OK;
}

template <class T, int NS, bool kLds>
int launch_prefactor(const PrefactorArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_prefactor<T, NS, kLds>;
    if (allow_big_lds(kern, lds_bytes)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(kThreads), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}

template <class T, int NS, bool kLds>
int launch_ipm(const IpmArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_ipm<T, NS, kLds>;
    if (allow_big_lds(kern, lds_bytes)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(kThreads), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}

template <class T, int NS, bool kLds, bool kBw>
int launch_kkt(const KktArgs<T>& a, size_t lds_bytes, void* stream)
{
    auto kern = k_kkt<T, NS, kLds, kBw>;
    if (allow_big_lds(kern, lds_bytes)) return QPX_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(kThreads), lds_bytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}

}  // namespace qpx

#include "qpx_api.inc"
