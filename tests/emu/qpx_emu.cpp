// tests/emu/qpx_emu.cpp -- libqpx_emu.so: the C ABI of include/qpx.h executed by host threads.
//
// TEST INFRASTRUCTURE ONLY (see tests/emu/qpx_platform.h).  Pointers are host pointers; the
// `stream` argument is ignored; workgroups run one after the other, each as QPX_EMU_THREADS
// (default 128 = two waves) fibers of the calling thread (pthreads with -DQPX_EMU_PTHREADS: the sanitizer
// drivers) over a heap-allocated "LDS".
#include <pthread.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "../../include/qpx.h"
#include "qpx_platform.h"  // the emulation header: defines QPX_PLATFORM_H, so the HIP one is skipped
#include "qpx_kernels.h"
#include "qpx_grid.h"
#include "qpx_tile.h"
#include "qpx_prefac.h"
#include "qpx_reduce.h"
#include "qpx_big.h"
#include "qpx_big_polish.h"

// Extra bytes behind the emulated LDS block.  The sanitizer build uses 0 so that an index one element past
// the size the launcher computed is already a reported overflow.
#ifndef QPX_EMU_LDS_SLACK
#define QPX_EMU_LDS_SLACK 64
#endif

namespace qpx {

template <int V> using Int = std::integral_constant<int, V>;
template <bool V> using Bool = std::integral_constant<bool, V>;

// QPX_EMU_LDS_BYTES shrinks the emulated LDS so that small problems exercise the
// "matrices stay in the HBM blob" code path.
inline size_t lds_budget_bytes()
{
    const char* e = std::getenv("QPX_EMU_LDS_BYTES");
    return e ? (size_t)std::atoll(e) : kMaxLdsBytes;
}

static int emu_threads()
{
    const char* e = std::getenv("QPX_EMU_THREADS");
    int nt = e ? std::atoi(e) : 128;
    if (nt < 64) nt = 64;
    return (nt / 64) * 64;
}

#ifdef QPX_EMU_PTHREADS
template <class Body> struct ThreadCtx {
    const Body* body;
    Block blk;
};

template <class Body> static void* thread_main(void* p)
{
    auto* c = static_cast<ThreadCtx<Body>*>(p);
    (*c->body)(c->blk);
    return nullptr;
}

// run `body(block)` for one workgroup
template <class Body> static void run_block(int nt, const Body& body)
{
    EmuShared sh;
    const int nw = nt / kWave;
    pthread_barrier_init(&sh.block_bar, nullptr, nt);
    std::vector<pthread_barrier_t> wb(nw);
    for (int w = 0; w < nw; ++w) pthread_barrier_init(&wb[w], nullptr, kWave);
    sh.wave_bar = wb.data();
    std::vector<unsigned long long> xchg(nt, 0);
    sh.xchg = xchg.data();
    std::vector<unsigned long long> xchg2(nt, 0);
    sh.xchg2 = xchg2.data();
    std::vector<double> xv((size_t)nt * 16, 0.0);
    sh.xv = xv.data();
    std::vector<ThreadCtx<Body>> ctx(nt);
    std::vector<pthread_t> th(nt);
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 1 << 20);
    for (int t = 0; t < nt; ++t) {
        ctx[t].body = &body;
        ctx[t].blk = Block{t, nt, &sh};
        pthread_create(&th[t], &attr, thread_main<Body>, &ctx[t]);
    }
    for (int t = 0; t < nt; ++t) pthread_join(th[t], nullptr);
    pthread_attr_destroy(&attr);
    for (int w = 0; w < nw; ++w) pthread_barrier_destroy(&wb[w]);
    pthread_barrier_destroy(&sh.block_bar);
}
#else
// ---------------------------------------------------------------------------------------------- fibers
// The GPU threads of one workgroup as user-level contexts of the calling host thread.  A context is a stack pointer:
// qpx_fiber_switch pushes the callee-saved registers of the System V x86-64 ABI, stores the stack pointer, loads the
// other context's and pops its registers (nothing in the kernel bodies changes MXCSR or the x87 control word).
extern "C" void qpx_fiber_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl qpx_fiber_switch
    .type qpx_fiber_switch,@function
qpx_fiber_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size qpx_fiber_switch, .-qpx_fiber_switch
    .section .note.GNU-stack,"",@progbits
    .text
)");

struct Fiber {
    void* sp = nullptr;
    FiberBar* waits = nullptr;     // the barrier this fiber sleeps at (nullptr: runnable)
    unsigned gen = 0;              // ... and the generation it arrived in
    bool done = false;
};
struct FiberSched {
    int nt = 0, cur = 0, ndone = 0;
    void* main_sp = nullptr;
    Fiber* f = nullptr;
    void (*entry)(FiberSched*, int) = nullptr;   // runs the body of fiber `tid`
    const void* body = nullptr;
    EmuShared* sh = nullptr;
};
static thread_local FiberSched* g_sched = nullptr;

// next runnable fiber after `cur` in round-robin order; -1 = every live fiber sleeps at a barrier nobody can open
static int fiber_pick(FiberSched* s)
{
    for (int k = 1; k <= s->nt; ++k) {
        const int t = (s->cur + k) % s->nt;
        Fiber& f = s->f[t];
        if (f.done) continue;
        if (f.waits && f.waits->gen == f.gen) continue;
        return t;
    }
    return -1;
}
static void fiber_deadlock(FiberSched* s)
{
    std::fprintf(stderr, "qpx_emu: dead-lock -- %d of %d GPU threads have finished and the others wait at barriers that "
                         "not every thread reaches (divergent __syncthreads / wave barrier in a kernel body)\n", s->ndone, s->nt);
    std::abort();
}
void fiber_wait(FiberSched* s, FiberBar* b, int parties)
{
    if (++b->count == parties) {       // last arrival: open the next generation and keep running
        b->count = 0;
        ++b->gen;
        return;
    }
    Fiber& me = s->f[s->cur];
    me.waits = b;
    me.gen = b->gen;
    const int t = fiber_pick(s);
    if (t < 0) fiber_deadlock(s);
    const int from = s->cur;
    s->cur = t;
    qpx_fiber_switch(&s->f[from].sp, s->f[t].sp);
    s->f[from].waits = nullptr;        // resumed: the generation has moved on
}
extern "C" void qpx_fiber_entry()
{
    FiberSched* s = g_sched;
    const int tid = s->cur;
    s->entry(s, tid);
    s->f[tid].done = true;
    ++s->ndone;
    const int t = fiber_pick(s);
    void* dead;
    if (t < 0) {
        if (s->ndone != s->nt) fiber_deadlock(s);
        qpx_fiber_switch(&dead, s->main_sp);          // the workgroup has finished
    } else {
        s->cur = t;
        qpx_fiber_switch(&dead, s->f[t].sp);
    }
    std::abort();                                      // a finished fiber is never resumed
}

// run `body(block)` for one workgroup
template <class Body> static void run_block(int nt, const Body& body)
{
    constexpr size_t kStack = 1 << 20;
    EmuShared sh;
    const int nw = nt / kWave;
    std::vector<FiberBar> wb(nw);
    sh.wave_bar = wb.data();
    std::vector<unsigned long long> xchg(nt, 0);
    sh.xchg = xchg.data();
    std::vector<unsigned long long> xchg2(nt, 0);
    sh.xchg2 = xchg2.data();
    std::vector<double> xv((size_t)nt * 16, 0.0);
    sh.xv = xv.data();
    // stacks: untouched pages of a fresh allocation cost nothing, so 1 MiB each as for the pthreads
    static thread_local unsigned char* stacks = nullptr;      // kept for the life of the host thread
    static thread_local size_t stacks_bytes = 0;
    if (stacks_bytes < (size_t)nt * kStack + 64) {
        std::free(stacks);
        stacks_bytes = (size_t)nt * kStack + 64;
        stacks = static_cast<unsigned char*>(std::malloc(stacks_bytes));
        if (!stacks) std::abort();
    }
    std::vector<Fiber> fib(nt);
    FiberSched sc;
    sc.nt = nt;
    sc.f = fib.data();
    sc.body = &body;
    sc.sh = &sh;
    sc.entry = [](FiberSched* s, int tid) { (*static_cast<const Body*>(s->body))(Block{tid, s->nt, s->sh}); };
    sh.sched = &sc;
    const uintptr_t base = ((uintptr_t)stacks + 63) & ~(uintptr_t)63;
    for (int t = 0; t < nt; ++t) {
        // initial frame: six zeroed callee-saved registers, the entry point as the return address of the first
        // switch, and a null return address above it (the entry function never returns); the stack pointer at
        // the entry is then 8 mod 16, as after a call
        uintptr_t top = (base + (size_t)(t + 1) * kStack) & ~(uintptr_t)15;
        void** p = reinterpret_cast<void**>(top);
        *--p = nullptr;
        *--p = reinterpret_cast<void*>(&qpx_fiber_entry);
        for (int r = 0; r < 6; ++r) *--p = nullptr;
        fib[t].sp = p;
    }
    FiberSched* outer = g_sched;
    g_sched = &sc;
    sc.cur = 0;
    qpx_fiber_switch(&sc.main_sp, fib[0].sp);
    g_sched = outer;
    if (sc.ndone != nt) fiber_deadlock(&sc);
}
#endif

template <class T, int NBL> int launch_sweep(const PrefactorArgs<T>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        T* base = reinterpret_cast<T*>(lds.data());
        run_block(256, [&](const Block& b) { sweep_body<T, NBL>(b, a, qp, base); });
    }
    return QPX_OK;
}
template <int NBN, bool kEq> int launch_prefac_tile(const PrefactorArgs<double>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        double* base = reinterpret_cast<double*>(lds.data());
        run_block(256, [&](const Block& b) { prefac_tile_body<NBN, kEq>(b, a, qp, base); });
    }
    return QPX_OK;
}
template <class T, int NBL, int NS> int launch_ipm_grid(const IpmArgs<T>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        T* base = reinterpret_cast<T*>(lds.data());
        run_block(256, [&](const Block& b) { ipm_grid_body<T, 16, NBL, NS>(b, a, qp, base); });
    }
    return QPX_OK;
}
template <class T, int NBL, int NS> int launch_ipm_grid8(const IpmArgs<T>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        T* base = reinterpret_cast<T*>(lds.data());
        run_block(64, [&](const Block& b) { ipm_grid_body<T, 8, NBL, NS>(b, a, qp, base); });
    }
    return QPX_OK;
}
template <int NBL, int NW, int NS, bool CH = false> int launch_ipm_tile(const IpmArgs<double>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        double* base = reinterpret_cast<double*>(lds.data());
        run_block(64 * NW, [&](const Block& b) { ipm_tile_body<NBL, NW, NS, CH>(b, a, qp, base); });
    }
    return QPX_OK;
}
template <int NBL, int NW, bool kBw, bool CH = false> int launch_kkt_tile(const KktArgs<double>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        double* base = reinterpret_cast<double*>(lds.data());
        run_block(64 * NW, [&](const Block& b) { kkt_tile_body<NBL, NW, kBw, CH>(b, a, qp, base); });
    }
    return QPX_OK;
}
template <class T, int NBL, bool kBw> int launch_kkt_grid(const KktArgs<T>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        T* base = reinterpret_cast<T*>(lds.data());
        run_block(256, [&](const Block& b) { kkt_grid_body<T, 16, NBL, kBw>(b, a, qp, base); });
    }
    return QPX_OK;
}

template <class T, int NBL> int launch_polish_grid(const PolishArgs<T>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        T* base = reinterpret_cast<T*>(lds.data());
        run_block(256, [&](const Block& b) { polish_grid_body<T, 16, NBL>(b, a, qp, base); });
    }
    return QPX_OK;
}
template <int NBL, int NW, bool CH> int launch_polish_tile(const PolishArgs<double>& a, size_t lds_bytes, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
        double* base = reinterpret_cast<double*>(lds.data());
        run_block(64 * NW, [&](const Block& b) { polish_mat_body<double, TileMat<NBL, NW, CH>>(b, a, qp, base); });
    }
    return QPX_OK;
}

// the large-QP family: grid (B, gy) of workgroups, one after the other
template <class F> static void big_grid(int B, int gy, int threads, size_t lds_bytes, const F& body)
{
    for (int y = 0; y < gy; ++y)
        for (int qp = 0; qp < B; ++qp) {
            std::vector<unsigned char> lds(lds_bytes + QPX_EMU_LDS_SLACK);
            unsigned char* base = lds.data();
            run_block(threads, [&](const Block& b) { body(b, qp, y, base); });
        }
}
template <class T> int launch_big_pack(const BigPackArgs<T>& a, int gy, void*)
{
    big_grid(a.B, gy, 256, 0, [&](const Block& b, int qp, int y, unsigned char*) { big_pack_body<T>(b, a, qp, y); });
    return QPX_OK;
}
template <class T> int launch_big_panel(const BigPanelArgs<T>& a, void*)
{
    big_grid(a.B, 1, 256, big_panel_lds_elems() * sizeof(T), [&](const Block& b, int qp, int, unsigned char* l) { big_panel_body<T>(b, a, qp, reinterpret_cast<T*>(l)); });
    return QPX_OK;
}
template <class T> int launch_big_gemm(const BigGemmArgs<T>& a, void*)
{
    big_grid(a.B, a.nti * a.ntj, 256, big_gemm2_lds_elems<T>(a.fuse != 0, a.mirror != 0) * sizeof(T), [&](const Block& b, int qp, int y, unsigned char* l) { big_gemm2_body<T, true>(b, a, qp, y, reinterpret_cast<T*>(l)); });
    return QPX_OK;
}
template <class T> int launch_big_trsv(const BigTrsvArgs<T>& a, void*)
{
    big_grid(a.B, 1, 64 * kTrsvNW, big_trsv_lds_elems(a.nb * kBB) * sizeof(T), [&](const Block& b, int qp, int, unsigned char* l) {
        if (a.nb > 8) big_trsv_body<T, true>(b, a, qp, reinterpret_cast<T*>(l));
        else big_trsv_body<T, false>(b, a, qp, reinterpret_cast<T*>(l));
    });
    return QPX_OK;
}
template <class T> int launch_big_gemv(const BigGemvArgs<T>& a, void*)
{
    const int outs = a.trans ? a.cols : a.rows;
    big_grid(a.B, (outs + kBB - 1) / kBB, 256, big_gemv_lds_elems(a.trans ? a.rows : a.cols) * sizeof(T), [&](const Block& b, int qp, int y, unsigned char* l) { big_gemv_body<T>(b, a, qp, y, reinterpret_cast<T*>(l)); });
    return QPX_OK;
}
template <class T> int launch_big_symv(const BigSymvArgs<T>& a, void*)
{
    big_grid(a.B, a.stage == 0 ? a.ld / kBB : 1, 256, big_symv_lds_elems() * sizeof(T), [&](const Block& b, int qp, int y, unsigned char* l) { big_symv_body<T>(b, a, qp, y, reinterpret_cast<T*>(l)); });
    return QPX_OK;
}
template <class T> int launch_big_vec(const BigVecArgs<T>& a, void*)
{
    big_grid(a.B, 1, 256, 0, [&](const Block& b, int qp, int, unsigned char*) { big_vec_body<T>(b, a, qp); });
    return QPX_OK;
}
template <class T> int launch_big_kkt(const BigKktArgs<T>& a, int gy, void*)
{
    big_grid(a.B, gy, 256, 0, [&](const Block& b, int qp, int y, unsigned char*) { big_kkt_body<T>(b, a, qp, y); });
    return QPX_OK;
}
template <class T> int launch_big_phase(const BigPhaseArgs<T>& a, void*)
{
    const int ns = big_pad(a.m) / kWave;
    big_grid(a.B, 1, 64, 0, [&](const Block& b, int qp, int, unsigned char*) {
        if (ns == 1) big_phase_body<T, 1>(b, a, qp);
        else if (ns == 2) big_phase_body<T, 2>(b, a, qp);
        else if (ns <= 4) big_phase_body<T, 4>(b, a, qp);
        else if (ns <= 8) big_phase_body<T, 8>(b, a, qp);
        else big_phase_body<T, 16>(b, a, qp);
    });
    return QPX_OK;
}

template <class T> int launch_big_solve(const BigSolveArgs<T>& a, void*)
{
    const int ns = big_pad(a.ph.m) / kWave;
    big_grid(a.t.B, 1, 64 * kTrsvNW, big_trsv_lds_elems(a.t.nb * kBB) * sizeof(T), [&](const Block& b, int qp, int, unsigned char* l) {
        T* lds = reinterpret_cast<T*>(l);
        if (ns == 1) big_solve_body<T, 1>(b, a, qp, lds);
        else if (ns == 2) big_solve_body<T, 2>(b, a, qp, lds);
        else if (ns <= 4) big_solve_body<T, 4>(b, a, qp, lds);
        else if (ns <= 8) big_solve_body<T, 8>(b, a, qp, lds);
        else big_solve_body<T, 16>(b, a, qp, lds);
    });
    return QPX_OK;
}
template <class T> int launch_big_polish(const BigPolishArgs<T>& a, void*)
{
    const BigLayout L = big_layout(a.n, a.m, a.q);
    big_grid(a.B, 1, 64 * kBigPolWaves, big_polish_lds_doubles(L.VP) * sizeof(double), [&](const Block& b, int qp, int, unsigned char* l) {
        big_polish_body<T>(b, a, qp, reinterpret_cast<double*>(l));
    });
    return QPX_OK;
}
template <class T> int launch_big_diag(const BigDiagArgs<T>& a, void*)
{
    const int ns = big_pad(a.ph.m) / kWave;
    big_grid(a.p.B, 1, 256, big_panel_lds_elems() * sizeof(T), [&](const Block& b, int qp, int, unsigned char* l) {
        T* lds = reinterpret_cast<T*>(l);
        if (ns == 1) big_diag_body<T, 1>(b, a, qp, lds);
        else if (ns == 2) big_diag_body<T, 2>(b, a, qp, lds);
        else if (ns <= 4) big_diag_body<T, 4>(b, a, qp, lds);
        else if (ns <= 8) big_diag_body<T, 8>(b, a, qp, lds);
        else big_diag_body<T, 16>(b, a, qp, lds);
    });
    return QPX_OK;
}

template <class T> int launch_batch_outer(const OuterArgs<T>& a, int tiles, void*)
{
    for (int ch = 0; ch < a.chunks; ++ch)
        for (int t = 0; t < tiles; ++t) {
            std::vector<T> lds((size_t)kOuterWaves * 256);
            run_block(64 * kOuterWaves, [&](const Block& b) { batch_outer_body<T>(b, a, t, ch, lds.data()); });
        }
    if (a.chunks > 1)
        for (int t = 0; t < tiles; ++t) run_block(256, [&](const Block& b) { batch_outer_sum_body<T>(b, a, t); });
    return QPX_OK;
}

template <class T> int launch_dense_solve(const DenseSolveArgs<T>& a, void*)
{
    for (int qp = 0; qp < a.B; ++qp) {
        std::vector<T> lds(dense_solve_lds_elems(a.k));
        run_block(256, [&](const Block& b) { dense_solve_body<T>(b, a, qp, lds.data()); });
    }
    return QPX_OK;
}

// side streams: launches are synchronous here, so the parts of a batch simply run one after the other
int stream_fork(void* caller, int nside, void** side, int, int = 0)
{
    for (int i = 0; i < nside; ++i) side[i] = caller;
    return QPX_OK;
}
int stream_join(void*, int, void* const*, int = 0) { return QPX_OK; }

}  // namespace qpx

#include "qpx_api.inc"

// test hook: force the "matrices in HBM" code path regardless of size
extern "C" int qpx_emu_marker(void) { return 1; }
