// qpx_hip_api.hip -- the C ABI of include/qpx.h (argument checks + dispatch); the kernels and
// launchers it calls are compiled in qpx_hip_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/qpx.h"
#include "qpx_launch.h"

namespace qpx {

// One pool of side streams + events per (host thread, device), created on first use and kept for the life of the
// thread.  Fork / join are event record + stream wait: stream-ordered, legal under stream capture, no host sync.
struct SidePool {
    // [0, kMaxSide): the streams of the parts of a batch beyond the first (which runs on the caller's);
    // [kMaxSide, kPoolSlots): one helper stream per part (R z' beside the factorisation), part i -> kMaxSide + i
    hipStream_t s[kPoolSlots];
    hipEvent_t fork[kPoolSlots], done[kPoolSlots];
    bool ok[kPoolSlots] = {};             // created on first use, slot by slot: a process holds the streams it uses (typically
                                          // one: the second part, or the helper), not seven -- streams share the device's four
                                          // hardware queues, and an idle stream that sits on the caller's queue serialises with it
};
constexpr int kMaxDev = 16;
static thread_local SidePool g_side[kMaxDev];

__global__ void k_stream_delay(long long ticks)
{
    // wall_clock64: constant 100 MHz; bounded above by the host (<= 10 ms), so it always terminates
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

int stream_fork(void* caller, int nside, void** side, int delay_us, int first)
{
    int dev = 0;
    if (nside < 1 || first < 0 || first + nside > kPoolSlots || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return QPX_ERR_LAUNCH;
    SidePool& p = g_side[dev];
    for (int i = first; i < first + nside; ++i) {
        if (p.ok[i]) continue;
        if (hipStreamCreateWithFlags(&p.s[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&p.fork[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p.done[i], hipEventDisableTiming) != hipSuccess)
            return QPX_ERR_LAUNCH;
        p.ok[i] = true;
    }
    // (one fork event per first slot: two parts of a batch fork their helper streams from different streams at once)
    if (hipEventRecord(p.fork[first], (hipStream_t)caller) != hipSuccess) return QPX_ERR_LAUNCH;
    for (int i = 0; i < nside; ++i) {
        if (hipStreamWaitEvent(p.s[first + i], p.fork[first], 0) != hipSuccess) return QPX_ERR_LAUNCH;
        if (delay_us > 0) {
            long long us = (long long)delay_us * (i + 1);
            if (us > 10000) us = 10000;
            hipLaunchKernelGGL(k_stream_delay, dim3(1), dim3(64), 0, p.s[first + i], us * 100);
        }
        side[i] = (void*)p.s[first + i];
    }
    return hipGetLastError() == hipSuccess ? QPX_OK : QPX_ERR_LAUNCH;
}

int stream_join(void* caller, int nside, void* const* side, int first)
{
    int dev = 0;
    if (nside < 1 || first < 0 || first + nside > kPoolSlots || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return QPX_ERR_LAUNCH;
    SidePool& p = g_side[dev];
    for (int i = 0; i < nside; ++i) {
        if (hipEventRecord(p.done[first + i], (hipStream_t)side[i]) != hipSuccess) return QPX_ERR_LAUNCH;
        if (hipStreamWaitEvent((hipStream_t)caller, p.done[first + i], 0) != hipSuccess) return QPX_ERR_LAUNCH;
    }
    return QPX_OK;
}

}  // namespace qpx

#include "qpx_api.inc"
