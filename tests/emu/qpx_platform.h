// tests/emu/qpx_platform.h -- HOST-THREAD EMULATION of qpth_amd/csrc/qpx_platform.h.
//
// TEST INFRASTRUCTURE ONLY.  (Shares the include guard QPX_PLATFORM_H with the HIP header so that
// including this file first makes qpx_kernels.h compile against the emulation.)  It lets the *same* kernel bodies (qpth_amd/csrc/qpx_kernels.h)
// run on the CPU -- one host context per GPU thread, barriers for __syncthreads and
// for the lock-step exchange behind wave shuffles -- so that indexing, control flow and
// barrier placement can be checked in the GPU-less build container.  Two execution modes:
//   * default: the GPU threads of a workgroup are FIBERS (user-level contexts with their own stacks) of the calling
//     host thread, switched at barriers by a round-robin scheduler -- no system calls, a barrier that not every
//     thread reaches is a reported dead-lock instead of a hang (a pthread barrier per shuffle cost ~15 minutes
//     of futex traffic per test run on an 8-core container);
//   * -DQPX_EMU_PTHREADS: one pthread per GPU thread with pthread barriers -- what ThreadSanitizer needs to see an
//     LDS exchange that no barrier orders (tests/emu/Makefile: tsan, asan).  It is compiled into tests/emu/_build/libqpx_emu.so, which only tests
// load; libqpx_hip.so never contains it and the qpth_amd package never falls back to it.
#ifndef QPX_PLATFORM_H
#define QPX_PLATFORM_H
#include <pthread.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#define QPX_DEV inline
#define QPX_HD inline
#define QPX_SCHED_FENCE() ((void)0)
#define QPX_LAUNDER_V(x) ((void)0)
#define QPX_LAUNDER_S(x) ((void)0)

namespace qpx {

constexpr int kWave = 64;

#ifndef QPX_EMU_PTHREADS
// A counting barrier for fibers: the last arrival opens the next generation, everybody else asks the scheduler for
// another runnable fiber until the generation has moved on.
struct FiberBar {
    int count = 0;
    unsigned gen = 0;
};
struct FiberSched;
void fiber_wait(FiberSched* s, FiberBar* b, int parties);      // qpx_emu.cpp
#endif

struct EmuShared {
#ifdef QPX_EMU_PTHREADS
    pthread_barrier_t block_bar;   // all threads of the workgroup
    pthread_barrier_t* wave_bar;   // one per wave
#else
    FiberBar block_bar;
    FiberBar* wave_bar;
    FiberSched* sched;
#endif
    unsigned long long* xchg;      // one 8-byte exchange slot per thread
    unsigned long long* xchg2;     // a second one (the B operand of the emulated MFMA)
    double* xv;                    // 16 doubles per thread (row_rank1: a whole register row in one round)
};

struct Block {
    int tid;
    int nt;
    EmuShared* sh;

    int lane() const { return tid & (kWave - 1); }
    int wave() const { return tid >> 6; }
    int nwaves() const { return nt >> 6; }
    int uniform(int v) const { return v; }
#ifdef QPX_EMU_PTHREADS
    void sync() const { pthread_barrier_wait(&sh->block_bar); }
    void sync_lds() const { sync(); }
    void wave_sync() const { pthread_barrier_wait(&sh->wave_bar[wave()]); }
#else
    void sync() const { fiber_wait(sh->sched, &sh->block_bar, nt); }
    void sync_lds() const { sync(); }
    void wave_sync() const { fiber_wait(sh->sched, &sh->wave_bar[wave()], kWave); }
#endif

    template <class T> T exchange(T v, int src_lane) const
    {
        unsigned long long bits = 0;
        std::memcpy(&bits, &v, sizeof(T));
        sh->xchg[tid] = bits;
        wave_sync();
        bits = sh->xchg[(tid & ~(kWave - 1)) | (src_lane & (kWave - 1))];
        wave_sync();
        T r;
        std::memcpy(&r, &bits, sizeof(T));
        return r;
    }
    float shfl_xor(float v, int mask) const { return exchange(v, lane() ^ mask); }
    double shfl_xor(double v, int mask) const { return exchange(v, lane() ^ mask); }
    int shfl_xor(int v, int mask) const { return exchange(v, lane() ^ mask); }
    float bcast(float v, int src) const { return exchange(v, src); }
    double bcast(double v, int src) const { return exchange(v, src); }
    int bcast(int v, int src) const { return exchange(v, src); }
    template <int K> double quad_bcast(double v) const { return exchange(v, (lane() & ~3) | K); }
    template <int K> float quad_bcast(float v) const { return exchange(v, (lane() & ~3) | K); }
    template <int MASK> double xor16(double v) const { return exchange(v, lane() ^ MASK); }
    template <int MASK> float xor16(float v) const { return exchange(v, lane() ^ MASK); }

    // DPP row_newbcast and the fused multiply-add through it (see the HIP header)
    template <int K> double row_bcast(double v) const { return exchange(v, (lane() & ~15) | K); }
    template <int K, int N> void row_rank1(double (&a)[N], double m) const
    {
        static_assert(N <= 16, "xv holds 16 values per thread");
        for (int j = 0; j < N; ++j) sh->xv[(size_t)tid * 16 + j] = a[j];
        wave_sync();
        const double* src = sh->xv + (size_t)((tid & ~15) | K) * 16;
        for (int j = 0; j < N; ++j) a[j] = std::fma(src[j], m, a[j]);
        wave_sync();
    }
    template <int GK> double grp_bcast(double v) const { return exchange(v, GK * 16 + (lane() & 15)); }

    bool any(bool v) const
    {
        sh->xchg[tid] = v ? 1ull : 0ull;
        wave_sync();
        unsigned long long r = 0;
        for (int l = 0; l < kWave; ++l) r |= sh->xchg[(tid & ~(kWave - 1)) | l];
        wave_sync();
        return r != 0;
    }

    template <int P> void prio() const {}

    // c += A B, A 16x4, B 4x16 (see the HIP header for the lane mapping); products are summed in
    // k order with fused multiply-adds -- the hardware's internal order is not documented, so GPU
    // and emulation agree to rounding, not bit for bit, once this is used
    void mfma16x16x4(double a, double b, double (&c)[4]) const
    {
        unsigned long long ba, bb;
        std::memcpy(&ba, &a, 8);
        std::memcpy(&bb, &b, 8);
        sh->xchg[tid] = ba;
        sh->xchg2[tid] = bb;
        wave_sync();
        const int base = tid & ~(kWave - 1), g = lane() >> 4, cc = lane() & 15;
        for (int r = 0; r < 4; ++r) {
            double acc = c[r];
            for (int k = 0; k < 4; ++k) {
                double av, bv;
                std::memcpy(&av, &sh->xchg[base + (k << 4) + (g + 4 * r)], 8);
                std::memcpy(&bv, &sh->xchg2[base + (k << 4) + cc], 8);
                acc = std::fma(av, bv, acc);
            }
            c[r] = acc;
        }
        wave_sync();
    }
    // four independent 4x4x4 products (see the HIP header): lane 16 h + 4 blk + j accumulates
    // sum_k A[lane 16 k + 4 blk + h] * B[lane 16 k + 4 blk + j]
    void mfma4x4x4(double a, double b, double& c) const
    {
        unsigned long long ba, bb;
        std::memcpy(&ba, &a, 8);
        std::memcpy(&bb, &b, 8);
        sh->xchg[tid] = ba;
        sh->xchg2[tid] = bb;
        wave_sync();
        const int base = tid & ~(kWave - 1), h = lane() >> 4, blk = (lane() >> 2) & 3, j = lane() & 3;
        double acc = c;
        for (int k = 0; k < 4; ++k) {
            double av, bv;
            std::memcpy(&av, &sh->xchg[base + (k << 4) + 4 * blk + h], 8);
            std::memcpy(&bv, &sh->xchg2[base + (k << 4) + 4 * blk + j], 8);
            acc = std::fma(av, bv, acc);
        }
        c = acc;
        wave_sync();
    }
    // f32 form: accumulator register r of lane group g is row 4 g + r
    void mfma16x16x4(float a, float b, float (&c)[4]) const
    {
        unsigned long long ba = 0, bb = 0;
        std::memcpy(&ba, &a, 4);
        std::memcpy(&bb, &b, 4);
        sh->xchg[tid] = ba;
        sh->xchg2[tid] = bb;
        wave_sync();
        const int base = tid & ~(kWave - 1), g = lane() >> 4, cc = lane() & 15;
        for (int r = 0; r < 4; ++r) {
            float acc = c[r];
            for (int k = 0; k < 4; ++k) {
                float av, bv;
                std::memcpy(&av, &sh->xchg[base + (k << 4) + (4 * g + r)], 4);
                std::memcpy(&bv, &sh->xchg2[base + (k << 4) + cc], 4);
                acc = std::fma(av, bv, acc);
            }
            c[r] = acc;
        }
        wave_sync();
    }
    static int mfma_row(double, int g, int r) { return g + 4 * r; }
    static int mfma_row(float, int g, int r) { return 4 * g + r; }
};

// workgroups run one after the other in the emulation: a plain add is the atomic
inline void atomic_add_(float* p, float v) { *p += v; }
inline void atomic_add_(double* p, double v) { *p += v; }

template <class T> struct GlobalRows {
    const T* base;
    int lane;
    GlobalRows(const T* b, int /*nelem*/, int l) : base(b), lane(l) {}
    T row(int r) const { return base[(size_t)r * kWave + lane]; }
};

template <class S> struct GlobalBuf {
    const S* base;
    long long nelem;
    GlobalBuf(const S* b, long long ne) : base(b), nelem(ne) {}
    double at(int off) const          // at or beyond nelem: zero (the raw buffer's range check); negative: a bug
    {
        if (off < 0) { std::fprintf(stderr, "GlobalBuf: negative offset %d\n", off); std::abort(); }
        return off >= nelem ? 0.0 : (double)base[off];
    }
    void at2(int off, double (&v)[2]) const { v[0] = at(off); v[1] = at(off + 1); }
};

template <class T> inline void ld4(const T* p, T (&v)[4]) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; }
template <class T> inline void st4(T* p, const T (&v)[4]) { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3]; }

template <class T> inline T fma_(T a, T b, T c) { return std::fma(a, b, c); }
inline float rcp_(float x) { return 1.0f / x; }
inline double rcp_(double x) { return 1.0 / x; }
inline float sqrt_(float x) { return std::sqrt(x); }
inline double sqrt_(double x) { return std::sqrt(x); }
inline float abs_(float x) { return std::fabs(x); }
inline double abs_(double x) { return std::fabs(x); }
inline bool finite_(float x) { return std::isfinite(x); }
inline bool finite_(double x) { return std::isfinite(x); }

}  // namespace qpx
#endif  // QPX_PLATFORM_H
