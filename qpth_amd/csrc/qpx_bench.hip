// qpx_bench.hip -- micro-benchmarks of the gfx950 primitives the kernels are built from
// (NOT part of libqpx_hip.so; built as libqpx_bench.so by `make bench`, driven by scripts/ubench.py).
// Every kernel runs single-wave workgroups and reports shader-clock cycles (clock64) per
// operation for wave 0 of every block.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "qpx_kernels.h"
#include "qpx_tile.h"

using namespace qpx;

#define BENCH_KERNEL(name) __global__ __launch_bounds__(64) void name(double* out, const double* in, int reps)

// 1. independent f64 FMAs: issue rate
BENCH_KERNEL(k_fma_tput)
{
    double a0 = in[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = in[64], c = in[65];
    long long w0 = wall_clock64();
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        }
    }
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[4096 + blockIdx.x] = double(t1 - t0) / (64.0 * reps);
        out[8192 + blockIdx.x] = double(w1 - w0) / (64.0 * reps);     // 100 MHz ticks per instruction
        out[12288 + blockIdx.x] = double(t1 - t0) / double(w1 - w0);  // clock64 ticks per 10 ns
    }
}
// 2. dependent f64 FMA chain: latency
BENCH_KERNEL(k_fma_lat)
{
    double a0 = in[threadIdx.x];
    const double m = in[64], c = in[65];
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 64; ++u) a0 = __builtin_fma(a0, m, c);
    }
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = a0;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (64.0 * reps);
}
// 3. dependent LDS read chain (pointer chasing): ds_read latency
BENCH_KERNEL(k_lds_lat)
{
    __shared__ int next[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) next[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 32; ++u) p = next[p];
    }
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = p;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (32.0 * reps);
}
// 4. LDS write by one lane group -> read by all lanes, dependent: publish/consume round trip
BENCH_KERNEL(k_lds_roundtrip)
{
    __shared__ double buf[256];
    const Block b{(int)threadIdx.x, 64};
    double v = in[threadIdx.x];
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if ((threadIdx.x >> 3) == (u & 7)) buf[(threadIdx.x & 7) + 8 * u] = v;
            b.wave_sync();
            v = v * 0.5 + buf[(threadIdx.x >> 3) + 8 * u];
        }
    }
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (16.0 * reps);
}
// 5. readlane -> fma chain (the substitution step)
BENCH_KERNEL(k_readlane_chain)
{
    const Block b{(int)threadIdx.x, 64};
    double x = in[threadIdx.x];
    const double l = in[64 + threadIdx.x] * 1e-3;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < 64; ++k) {
            const double yk = b.bcast(x, k);
            x = __builtin_fma(-l, yk, x);
        }
    }
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (64.0 * reps);
}
// 6. reciprocal: rcp_ (estimate + 2 Newton) and full division, dependent chains
BENCH_KERNEL(k_rcp_lat)
{
    double x = in[threadIdx.x] + 1.5, y = x;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = rcp_(x) + 1.25;
    }
    long long t1 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) y = 1.0 / y + 1.25;
    }
    long long t2 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = x + y;
    if (threadIdx.x == 0) {
        out[4096 + blockIdx.x] = double(t1 - t0) / (16.0 * reps);
        out[8192 + blockIdx.x] = double(t2 - t1) / (16.0 * reps);
    }
}
// 7. ds_bpermute (shfl_xor) dependent chain and 8-wide independent
BENCH_KERNEL(k_bpermute)
{
    const Block b{(int)threadIdx.x, 64};
    double x = in[threadIdx.x];
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 6; ++u) x += b.shfl_xor(x, 1 << u);
    }
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = x;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (6.0 * reps);
}

// 10. v_mfma_f64_16x16x4_f64: NACC independent accumulators issued round-robin (throughput at
// NACC = 8, dependent-chain latency at NACC = 1); and the same rank-4 update of a 16x16 tile done
// with 16 vector FMAs per lane (operands in registers) for comparison
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int NACC> __global__ __launch_bounds__(64) void k_mfma_f64(double* out, const double* in, int reps)
{
    d4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4_t{in[threadIdx.x], in[threadIdx.x + 64], 0.0, 1.0};
    const double a = in[threadIdx.x + 128] * 1e-3, bb = in[threadIdx.x + 192] * 1e-3;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[i], 0, 0, 0);
    }
    double sum = 0;
    for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (4.0 * NACC * reps);
}
// 10b. v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction, ONE accumulator register per lane:
// 256 multiply-adds against the 1024 of the 16x16x4 form): what a tile row with four real rows of sixteen would use
template <int NACC> __global__ __launch_bounds__(64) void k_mfma_f64_4x4(double* out, const double* in, int reps)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = in[threadIdx.x] + i;
    const double a = in[threadIdx.x + 128] * 1e-3, bb = in[threadIdx.x + 192] * 1e-3;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, acc[i], 0, 0, 0);
    }
    double sum = 0;
    for (int i = 0; i < NACC; ++i) sum += acc[i];
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (4.0 * NACC * reps);
}
// 10c. the lane layout of v_mfma_f64_4x4x4_4b_f64, decoded with powers of two: out[x * 64 + lane] for x = 0 .. 17:
//   x = 16: A = 2^(lane % 16), B = 1          -> D = sum over k of A: which A lanes feed each D lane
//   x = 17: A = 1, B = 2^(lane % 16)          -> which B lanes feed each D lane
//   x < 16: A = 2^(lane % 16), B = 1 in lanes with lane % 16 == x only (0 elsewhere) -> the A lane that meets B lane x
__global__ __launch_bounds__(64) void k_mfma4_layout(double* out, const double* in, int reps)
{
    const int lane = threadIdx.x;
    const double p2 = double(1 << (lane & 15));
    for (int x = 0; x < 18; ++x) {
        const double a = x == 17 ? 1.0 : p2;
        const double b = x == 16 ? 1.0 : (x == 17 ? p2 : ((lane & 15) == x ? 1.0 : 0.0));
        out[x * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    }
    // x = 18: blocks: A = 1 in block (lane / 16) == 0 only, B = 1 -> which D lanes see block 0's A
    out[18 * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(lane < 16 ? 1.0 : 0.0, 1.0, 0.0, 0, 0, 0);
}
template <int NACC> __global__ __launch_bounds__(64) void k_vfma_tile(double* out, const double* in, int reps)
{
    double acc[NACC][4];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = in[threadIdx.x + 64 * r];
    double l[4][4], v[4];
    for (int r = 0; r < 4; ++r) {
        v[r] = in[threadIdx.x + 256 + r] * 1e-3;
        for (int k = 0; k < 4; ++k) l[r][k] = in[threadIdx.x + 300 + 4 * r + k] * 1e-3;
    }
    long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][r] = __builtin_fma(l[r][k], v[k], acc[i][r]);
    }
    double sum = 0;
    for (int i = 0; i < NACC; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    long long t1 = clock64();
    out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) out[4096 + blockIdx.x] = double(t1 - t0) / (4.0 * NACC * reps);
}

// The 16 x 16 pivot block of the sixteen-column panels (TileMat::pivot16 x 16, registers only), alone in its
// workgroup (PARTNER = 0) or with a second wave that lands on the same SIMD (a workgroup's waves are dealt round-robin
// over the four SIMDs: wave 4 joins wave 0) and streams independent f64 MFMAs (1), f64 vector FMAs (2), or f32 FMAs
// (3) for as long as wave 0 works.  Reports shader-clock ticks per pivot block for wave 0.
template <int PARTNER> __global__ __launch_bounds__(PARTNER ? 320 : 64) void k_pivot_block(double* out, const double* in, int reps)
{
    typedef TileMat<7, 4> TM;
    __shared__ volatile int done;
    __shared__ double lines[5 * 32];                  // the pivot-row line of every wave (TileMat::pivot16)
    const Block b{(int)(threadIdx.x & 63), 64};
    const int wave = threadIdx.x >> 6;
    double* line = lines + 32 * wave;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave == 0) {
        const TM::Pos p(b);
        long long t0 = clock64();
        double sum = 0;
        for (int r = 0; r < reps; ++r) {
            double a[4];
            // a diagonally dominant symmetric block: row c, columns 4 g .. 4 g + 3
            for (int j = 0; j < 4; ++j) a[j] = (p.c == 4 * p.g + j ? 20.0 : 0.0) + in[(p.c * 16 + 4 * p.g + j + r) & 1023] + in[((4 * p.g + j) * 16 + p.c + r) & 1023];
            double dg = 20.0 + 2 * in[(p.c * 17 + r) & 1023], myr = 1.0;
            double vn_a = b.template grp_bcast<0>(a[0]);        // column 0 in every lane group (TileMat::pivot_block)
            TM::pivot16<0>(b, p, a, dg, myr, vn_a); TM::pivot16<1>(b, p, a, dg, myr, vn_a); TM::pivot16<2>(b, p, a, dg, myr, vn_a); TM::pivot16<3>(b, p, a, dg, myr, vn_a);
            TM::pivot16<4>(b, p, a, dg, myr, vn_a); TM::pivot16<5>(b, p, a, dg, myr, vn_a); TM::pivot16<6>(b, p, a, dg, myr, vn_a); TM::pivot16<7>(b, p, a, dg, myr, vn_a);
            TM::pivot16<8>(b, p, a, dg, myr, vn_a); TM::pivot16<9>(b, p, a, dg, myr, vn_a); TM::pivot16<10>(b, p, a, dg, myr, vn_a); TM::pivot16<11>(b, p, a, dg, myr, vn_a);
            TM::pivot16<12>(b, p, a, dg, myr, vn_a); TM::pivot16<13>(b, p, a, dg, myr, vn_a); TM::pivot16<14>(b, p, a, dg, myr, vn_a); TM::pivot16<15>(b, p, a, dg, myr, vn_a);
            sum += a[0] + a[1] + a[2] + a[3] + myr;
        }
        long long t1 = clock64();
        out[20480 + (blockIdx.x & 63) * 64 + threadIdx.x] = sum;
        if (threadIdx.x == 0) {
            out[4096 + blockIdx.x] = double(t1 - t0) / reps;
            done = 1;
        }
    } else if (PARTNER && wave == 4) {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{in[threadIdx.x & 63], 0.0, 1.0, 2.0};
        double x0 = in[threadIdx.x & 63], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        float f0 = (float)x0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
        const double m = in[64] * 1e-3, c = in[65];
        long long n = 0;
        while (!done) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (PARTNER == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(m, c, acc[i], 0, 0, 0);
                } else if (PARTNER == 2) {
                    x0 = __builtin_fma(x0, m, c); x1 = __builtin_fma(x1, m, c); x2 = __builtin_fma(x2, m, c); x3 = __builtin_fma(x3, m, c);
                } else {
                    f0 = __builtin_fmaf(f0, (float)m, (float)c); f1 = __builtin_fmaf(f1, (float)m, (float)c);
                    f2 = __builtin_fmaf(f2, (float)m, (float)c); f3 = __builtin_fmaf(f3, (float)m, (float)c);
                }
            }
            ++n;
        }
        out[16384 + (threadIdx.x & 63)] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3;
        if ((threadIdx.x & 63) == 0) out[8192 + blockIdx.x] = (double)n;
    }
}

// accuracy of rcp_ against IEEE division over many magnitudes
__global__ void k_rcp_err(double* out, const double* in, int reps)
{
    double worst = 0, worst_raw = 0;
    for (int r = 0; r < reps; ++r) {
        const double x = ldexp(in[(threadIdx.x + 64 * r) & 4095], (r * 37 + (int)threadIdx.x) % 200 - 100);
        const double ex = 1.0 / x;
        const double e1 = __builtin_fabs(rcp_(x) - ex) / __builtin_fabs(ex);
        const double e0 = __builtin_fabs(__builtin_amdgcn_rcp(x) - ex) / __builtin_fabs(ex);
        worst = e1 > worst ? e1 : worst;
        worst_raw = e0 > worst_raw ? e0 : worst_raw;
    }
    out[4096 + threadIdx.x] = worst;
    out[8192 + threadIdx.x] = worst_raw;
}

// 40. Where do the waves of a workgroup land?  256-thread workgroups shaped like the tile loop kernel (two per CU:
// launch bounds (256, 2) + ~70 KB of dynamic LDS); every wave records HW_ID (SIMD, CU, SE) and XCC_ID, then the
// workgroup idles long enough for the whole grid to be co-resident.  out[4096 + 4 * (4 * block + wave) + {0, 1, 2}].
__global__ __launch_bounds__(256, 2) void k_simd_probe(double* out, const double* in, int reps)
{
    extern __shared__ double probe_lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int wave = threadIdx.x >> 6;
    probe_lds[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    double x = probe_lds[(threadIdx.x + 1) & 255];
    const long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) x = __builtin_fma(x, 0.999, 0.001);
    if ((threadIdx.x & 63) == 0) {
        double* o = out + 4096 + 4 * (4 * (size_t)blockIdx.x + wave);
        o[0] = (double)hw; o[1] = (double)xcc; o[2] = (double)t0; o[3] = x;
    }
}

// 41 .. 43. The pivot block beside (41) a second wave on the same SIMD that runs the same pivot blocks, (42) an MFMA
// stream at s_setprio 0 while the pivot wave runs at s_setprio 3, (43) an MFMA stream with gaps (4 MFMAs, then as
// many idle cycles: a 50 % duty cycle)
template <int MODE> __global__ __launch_bounds__(320) void k_pivot_pair(double* out, const double* in, int reps)
{
    typedef TileMat<7, 4> TM;
    __shared__ volatile int done;
    __shared__ double lines[5 * 32];                  // the pivot-row line of every wave (TileMat::pivot16)
    const Block b{(int)(threadIdx.x & 63), 64};
    const int wave = threadIdx.x >> 6;
    double* line = lines + 32 * wave;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave == 0 || (MODE == 1 && wave == 4)) {
        const TM::Pos p(b);
        if (MODE == 2 || MODE == 3) __builtin_amdgcn_s_setprio(3);
        long long t0 = clock64();
        double sum = 0;
        for (int r = 0; r < reps; ++r) {
            double a[4];
            for (int j = 0; j < 4; ++j) a[j] = (p.c == 4 * p.g + j ? 20.0 : 0.0) + in[(p.c * 16 + 4 * p.g + j + r) & 1023] + in[((4 * p.g + j) * 16 + p.c + r) & 1023];
            double dg = 20.0 + 2 * in[(p.c * 17 + r) & 1023], myr = 1.0;
            double vn_a = b.template grp_bcast<0>(a[0]);        // column 0 in every lane group (TileMat::pivot_block)
            TM::pivot16<0>(b, p, a, dg, myr, vn_a); TM::pivot16<1>(b, p, a, dg, myr, vn_a); TM::pivot16<2>(b, p, a, dg, myr, vn_a); TM::pivot16<3>(b, p, a, dg, myr, vn_a);
            TM::pivot16<4>(b, p, a, dg, myr, vn_a); TM::pivot16<5>(b, p, a, dg, myr, vn_a); TM::pivot16<6>(b, p, a, dg, myr, vn_a); TM::pivot16<7>(b, p, a, dg, myr, vn_a);
            TM::pivot16<8>(b, p, a, dg, myr, vn_a); TM::pivot16<9>(b, p, a, dg, myr, vn_a); TM::pivot16<10>(b, p, a, dg, myr, vn_a); TM::pivot16<11>(b, p, a, dg, myr, vn_a);
            TM::pivot16<12>(b, p, a, dg, myr, vn_a); TM::pivot16<13>(b, p, a, dg, myr, vn_a); TM::pivot16<14>(b, p, a, dg, myr, vn_a); TM::pivot16<15>(b, p, a, dg, myr, vn_a);
            sum += a[0] + a[1] + a[2] + a[3] + myr;
        }
        long long t1 = clock64();
        out[20480 + (blockIdx.x & 63) * 64 + (threadIdx.x & 63)] = sum;
        if ((threadIdx.x & 63) == 0) {
            out[(wave == 0 ? 4096 : 12288) + blockIdx.x] = double(t1 - t0) / reps;
            if (wave == 0) done = 1;
        }
    } else if (MODE != 1 && wave == 4) {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{in[threadIdx.x & 63], 0.0, 1.0, 2.0};
        const double m = in[64] * 1e-3, c = in[65];
        long long n = 0;
        __builtin_amdgcn_s_setprio(0);
        while (!done) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(m, c, acc[i], 0, 0, 0);
            if (MODE == 3) __builtin_amdgcn_s_sleep(5);     // ~320 cycles
            ++n;
        }
        out[16384 + (threadIdx.x & 63)] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        if ((threadIdx.x & 63) == 0) out[8192 + blockIdx.x] = (double)n;
    }
}

// 50. Streams `m` doubles per workgroup through raw_buffer_load_b64 rows (the GlobalRows pattern of the tile loads),
// 8 B per lane and load, every byte read once: calibration of rocprofv3's FETCH_SIZE for that access width.
__global__ __launch_bounds__(256) void k_stream_b64(double* out, const double* in, int reps, int m)
{
    const Block b{(int)threadIdx.x, 256};
    const int rows = m / 64;                       // rows of 64 doubles per workgroup
    const GlobalRows<double> g(in + (size_t)blockIdx.x * m, m, b.lane());
    double acc = 0;
    for (int r = b.wave(); r < rows; r += 4) acc += g.row(r);
    if (acc == 12345.678) out[threadIdx.x] = acc;
}
// 51. the same bytes through 16-B-per-lane global loads (the access width the guide's x2 correction is calibrated for)
__global__ __launch_bounds__(256) void k_stream_b128(double* out, const double* in, int reps, int m)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* src = reinterpret_cast<const d2*>(in + (size_t)blockIdx.x * m);
    double acc = 0;
    for (int i = threadIdx.x; i < m / 2; i += 256) { const d2 v = src[i]; acc += v[0] + v[1]; }
    if (acc == 12345.678) out[threadIdx.x] = acc;
}

extern "C" int qpx_bench_ptr(int which, int blocks, int reps, int m, double* out, const double* in, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    switch (which) {
    case 40: {
        static bool once = false;
        if (!once) { once = true; hipFuncSetAttribute((const void*)k_simd_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); }
        hipLaunchKernelGGL(k_simd_probe, dim3(blocks), dim3(256), 72 * 1024, s, out, in, reps);
        break;
    }
    case 41: hipLaunchKernelGGL(k_pivot_pair<1>, dim3(blocks), dim3(320), 0, s, out, in, reps); break;
    case 42: hipLaunchKernelGGL(k_pivot_pair<2>, dim3(blocks), dim3(320), 0, s, out, in, reps); break;
    case 43: hipLaunchKernelGGL(k_pivot_pair<3>, dim3(blocks), dim3(320), 0, s, out, in, reps); break;
    case 50: hipLaunchKernelGGL(k_stream_b64, dim3(blocks), dim3(256), 0, s, out, in, reps, m); break;
    case 51: hipLaunchKernelGGL(k_stream_b128, dim3(blocks), dim3(256), 0, s, out, in, reps, m); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

extern "C" int qpx_bench(int which, int blocks, int reps, int m, double* out, const double* in, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    switch (which) {
    case 1: hipLaunchKernelGGL(k_fma_tput, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 2: hipLaunchKernelGGL(k_fma_lat, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 3: hipLaunchKernelGGL(k_lds_lat, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 4: hipLaunchKernelGGL(k_lds_roundtrip, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 5: hipLaunchKernelGGL(k_readlane_chain, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 6: hipLaunchKernelGGL(k_rcp_lat, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 9: hipLaunchKernelGGL(k_rcp_err, dim3(1), dim3(64), 0, s, out, in, reps); break;
    case 7: hipLaunchKernelGGL(k_bpermute, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 20: hipLaunchKernelGGL(k_mfma_f64<8>, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 21: hipLaunchKernelGGL(k_mfma_f64<1>, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 22: hipLaunchKernelGGL(k_mfma_f64<2>, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 23: hipLaunchKernelGGL(k_vfma_tile<8>, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 26: hipLaunchKernelGGL(k_mfma4_layout, dim3(1), dim3(64), 0, s, out, in, reps); break;
    case 24: hipLaunchKernelGGL(k_mfma_f64_4x4<8>, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 25: hipLaunchKernelGGL(k_mfma_f64_4x4<1>, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 30: hipLaunchKernelGGL(k_pivot_block<0>, dim3(blocks), dim3(64), 0, s, out, in, reps); break;
    case 31: hipLaunchKernelGGL(k_pivot_block<1>, dim3(blocks), dim3(320), 0, s, out, in, reps); break;
    case 32: hipLaunchKernelGGL(k_pivot_block<2>, dim3(blocks), dim3(320), 0, s, out, in, reps); break;
    case 33: hipLaunchKernelGGL(k_pivot_block<3>, dim3(blocks), dim3(320), 0, s, out, in, reps); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
