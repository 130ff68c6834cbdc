// qpx_hip_api.hip -- the C ABI of include/qpx.h (argument checks + dispatch); the kernels and
// launchers it calls are compiled in qpx_hip_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "../../include/qpx.h"
#include "qpx_launch.h"

namespace qpx {

// One pool of side streams + events per (host thread, device), created on first use and kept for the life of the
// thread.  Fork / join are event record + stream wait: stream-ordered, legal under stream capture, no host sync (the one
// wait of the library is the check of a new side stream against the caller's, side_slot below).
struct SidePool {
    // [0, kMaxSide): the streams of the parts of a batch beyond the first (which runs on the caller's);
    // [kMaxSide, kPoolSlots): one helper stream per part (R z' beside the factorisation), part i -> kMaxSide + i
    hipStream_t s[kPoolSlots];
    hipEvent_t fork[kPoolSlots], done[kPoolSlots];
    // the caller's streams a slot has been checked against (runs_beside), most recent kSeen of them: a caller that
    // alternates between a few streams is probed once per stream, not at every switch
    static constexpr int kSeen = 8;
    void* beside[kPoolSlots][kSeen] = {};
    int nseen[kPoolSlots] = {};
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

// HIP deals streams to the device's (four) hardware queues by a policy the API does not expose.  A side stream that lands on
// the CALLER's queue does not run beside it: the two parts of a batch then execute one after the other -- twice the latency
// chain of one part, 24.9 instead of 11.5 ms per step at C4 in a process that had used six other streams before
// (profiles/r05p_stream_clash.txt).  So a slot is CHECKED once against the caller's stream it is used with (and again when that
// stream changes): one 100-us delay kernel on each of the two streams between two timed events -- ~0.1 ms if they run side by
// side, ~0.2 ms if they share a queue.  That is the one place where the library waits on the host (~0.3 ms, once per host thread,
// device and caller stream); it is skipped while the caller's stream is being captured into a graph.
static bool runs_beside(hipStream_t caller, hipStream_t side)
{
    // the first launch of the delay kernel in a process loads its code object: not inside the timed pair (a cold first
    // candidate would be rejected for that alone)
    static thread_local bool warmed = false;
    if (!warmed) {
        hipLaunchKernelGGL(k_stream_delay, dim3(1), dim3(64), 0, side, 0LL);
        (void)hipStreamSynchronize(side);
        warmed = true;
    }
    hipEvent_t e0, e1, es;
    if (hipEventCreate(&e0) != hipSuccess) return true;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return true; }
    if (hipEventCreateWithFlags(&es, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return true; }
    float ms = 0.f;
    bool ok = hipEventRecord(e0, caller) == hipSuccess && hipStreamWaitEvent(side, e0, 0) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_stream_delay, dim3(1), dim3(64), 0, side, 100LL * 100);
        hipLaunchKernelGGL(k_stream_delay, dim3(1), dim3(64), 0, caller, 100LL * 100);
        ok = hipEventRecord(es, side) == hipSuccess && hipStreamWaitEvent(caller, es, 0) == hipSuccess &&
             hipEventRecord(e1, caller) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
             hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(es);
    return !ok || ms < 0.16f;             // (a failed measurement decides nothing)
}

// make slot i of the pool a stream that runs beside `caller`: up to four candidates (one per hardware queue); the rejected ones
// stay alive until the choice is made, so that the next candidate is dealt another queue
static int side_slot(SidePool& p, int i, hipStream_t caller)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(caller, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (!p.ok[i]) {
        if (hipStreamCreateWithFlags(&p.s[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&p.fork[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p.done[i], hipEventDisableTiming) != hipSuccess)
            return QPX_ERR_LAUNCH;
        p.ok[i] = true;
        p.nseen[i] = 0;
    }
    if (capturing) return QPX_OK;
    for (int k = 0; k < p.nseen[i]; ++k)
        if (p.beside[i][k] == (void*)caller) return QPX_OK;
    static const bool no_probe = std::getenv("QPX_NO_STREAM_PROBE") != nullptr;
    if (no_probe) return QPX_OK;
    if (p.nseen[i] > 0 && runs_beside(caller, p.s[i])) {
        // the slot's stream runs beside this stream of the caller too: one more verdict to remember
        if (p.nseen[i] == SidePool::kSeen) {
            for (int k = 1; k < SidePool::kSeen; ++k) p.beside[i][k - 1] = p.beside[i][k];
            p.nseen[i] = SidePool::kSeen - 1;
        }
        p.beside[i][p.nseen[i]++] = (void*)caller;
        return QPX_OK;
    }
    // a fresh slot, or one whose stream shares a hardware queue with this caller stream: candidates until one runs beside
    // it (the verdicts about the replaced stream go with it)
    p.nseen[i] = 0;
    hipStream_t rejected[3];
    int nrej = 0;
    while (!runs_beside(caller, p.s[i]) && nrej < 3) {
        hipStream_t next;
        if (hipStreamCreateWithFlags(&next, hipStreamNonBlocking) != hipSuccess) break;
        rejected[nrej++] = p.s[i];
        p.s[i] = next;
    }
    for (int r = 0; r < nrej; ++r) (void)hipStreamDestroy(rejected[r]);
    p.beside[i][p.nseen[i]++] = (void*)caller;
    return QPX_OK;
}

int stream_fork(void* caller, int nside, void** side, int delay_us, int first)
{
    int dev = 0;
    if (nside < 1 || first < 0 || first + nside > kPoolSlots || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return QPX_ERR_LAUNCH;
    SidePool& p = g_side[dev];
    for (int i = first; i < first + nside; ++i)
        if (const int e = side_slot(p, i, (hipStream_t)caller)) return e;
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
