// qpx_platform.h -- gfx950 (CDNA4) execution primitives used by the qpx kernels.
//
// The kernel bodies in qpx_kernels.h are written against this tiny vocabulary (wave64,
// workgroup barrier, intra-wave LDS ordering point, wave shuffles).  The test tree carries a
// second header of the same name (tests/emu/qpx_platform.h) that runs the SAME kernel bodies
// on host threads so that indexing/logic can be checked without a GPU and under
// ThreadSanitizer; that emulation is test infrastructure and is never linked into
// libqpx_hip.so.
#ifndef QPX_PLATFORM_H
#define QPX_PLATFORM_H
#include <hip/hip_runtime.h>

#define QPX_DEV __device__ __forceinline__
#define QPX_HD __host__ __device__ __forceinline__
// keep the compiler's scheduler from moving instructions across this point (order of MFMA groups and LDS writes)
#ifndef QPX_SCHED_FENCE
#define QPX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// Make a value opaque to the optimiser at this point (vector / scalar register).  The tile kernels re-derive their LDS
// addresses from laundered lane coordinates at the top of every panel: otherwise the compiler hoists some sixty
// loop-invariant address registers and branch conditions out of the interior-point loop, keeps them alive across
// the whole kernel and spills them (measured: 43 spilled VGPRs in the chain-wave form).
#define QPX_LAUNDER_V(x) asm volatile("" : "+v"(x))
#define QPX_LAUNDER_S(x) asm volatile("" : "+s"(x))

namespace qpx {

constexpr int kWave = 64;  // CDNA wavefront width

struct Block {
    int tid;  // thread index in the workgroup
    int nt;   // workgroup size (multiple of 64)

    QPX_DEV int lane() const { return tid & (kWave - 1); }
    QPX_DEV int wave() const { return tid >> 6; }
    QPX_DEV int nwaves() const { return nt >> 6; }
    // a value every lane of the wave holds, moved to a scalar register so that branches on it are
    // scalar branches (the compiler cannot see that tid >> 6 is wave-uniform)
    QPX_DEV int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }

    // workgroup barrier; LDS and global writes of the workgroup made before it are visible after.
    // The explicit wait is load-bearing: hipcc (ROCm 7.2) dropped the `s_waitcnt lgkmcnt(0)`
    // that belongs to __syncthreads() on the back-edge of the Cholesky column loop (every other
    // barrier of the kernel kept it), so a ds_write of one wave could still be in flight when
    // another wave read the element after the barrier -- seen on MI355X as ~1 QP in 4096
    // spuriously flagged "not SPD".  Inline asm is invisible to the pass that removes the wait.
    QPX_DEV void sync() const
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // workgroup barrier for kernels whose waves exchange data through LDS ONLY: global loads issued before it stay in
    // flight across it (sync() drains them).  big_trsv_body prefetches a block step ahead under its barriers.
    QPX_DEV void sync_lds() const
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ordering point for LDS traffic between lanes of ONE wave.  A wave's DS instructions
    // execute in issue order, so only the compiler must be kept from moving them.
    QPX_DEV void wave_sync() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    QPX_DEV float shfl_xor(float v, int mask) const { return __shfl_xor(v, mask, kWave); }
    QPX_DEV double shfl_xor(double v, int mask) const { return __shfl_xor(v, mask, kWave); }
    QPX_DEV int shfl_xor(int v, int mask) const { return __shfl_xor(v, mask, kWave); }

    // value of `v` held by lane `src` (src must be wave-uniform): v_readlane, no LDS traffic
    QPX_DEV float bcast(float v, int src) const
    {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
    }
    QPX_DEV double bcast(double v, int src) const
    {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
        const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    QPX_DEV int bcast(int v, int src) const { return __builtin_amdgcn_readlane(v, src); }

    // value of `v` held by lane (lane ^ MASK), MASK in {1, 2, 7, 15}: partners inside a row of 16
    // lanes, done with DPP moves (quad_perm / row_half_mirror / row_mirror) -- no LDS, ~3 VALU
    // issues for a double.  The four masks in sequence sum a value over the 16 lanes of a row.
    template <int MASK> QPX_DEV double xor16(double v) const
    {
        static_assert(MASK == 1 || MASK == 2 || MASK == 7 || MASK == 15, "DPP row pattern");
        constexpr int ctrl = MASK == 1 ? 0xB1 : (MASK == 2 ? 0x4E : (MASK == 7 ? 0x141 : 0x140));
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), ctrl, 0xF, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), ctrl, 0xF, 0xF, false);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    template <int MASK> QPX_DEV float xor16(float v) const
    {
        static_assert(MASK == 1 || MASK == 2 || MASK == 7 || MASK == 15, "DPP row pattern");
        constexpr int ctrl = MASK == 1 ? 0xB1 : (MASK == 2 ? 0x4E : (MASK == 7 ? 0x141 : 0x140));
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false));
    }

    // value of `v` held by lane K of this lane's QUAD (lanes 4 j .. 4 j + 3): DPP quad_perm [K, K, K, K]
    template <int K> QPX_DEV double quad_bcast(double v) const
    {
        static_assert(K >= 0 && K < 4, "quad of 4 lanes");
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), K * 0x55, 0xF, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), K * 0x55, 0xF, 0xF, false);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    template <int K> QPX_DEV float quad_bcast(float v) const
    {
        static_assert(K >= 0 && K < 4, "quad of 4 lanes");
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xF, 0xF, false));
    }

    // value of `v` held by lane K of this lane's row of 16 lanes: DPP row_newbcast, the one DPP pattern the
    // f64 ALU of gfx90a+ takes (v_mov_b64_dpp; the compiler folds it into v_rcp_f64_dpp and friends)
    template <int K> QPX_DEV double row_bcast(double v) const
    {
        return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xF, 0xF, true);
    }
    // a[j] += (a[j] of lane K of this row of 16) * m for every j: ONE instruction per element
    // (v_fmac_f64_dpp, src0 through the row broadcast).  The compiler does not form it from update_dpp + fma
    // (it keeps a v_mov_b64_dpp per element), hence the assembly.  Hazard "VALU writes a VGPR -> DPP reads it: 2
    // wait states" is not tracked through inline assembly: the s_nop in front of the first element covers the
    // registers written just before the group; inside the group every element reads a register of its own.
    template <int K, int N> QPX_DEV void row_rank1(double (&a)[N], double m) const
    {
        static_assert(K >= 0 && K < 16, "row of 16 lanes");
        bool first = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {
#define QPX_FMAC_DPP(KK)                                                                                          \
    if constexpr (K == KK) {                                                                                      \
        if (first)                                                                                                \
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:" #KK " row_mask:0xf bank_mask:0xf" \
                         : "+v"(a[j])                                                                             \
                         : "v"(m));                                                                               \
        else                                                                                                      \
            asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:" #KK " row_mask:0xf bank_mask:0xf"              \
                         : "+v"(a[j])                                                                             \
                         : "v"(m));                                                                               \
    }
            QPX_FMAC_DPP(0) QPX_FMAC_DPP(1) QPX_FMAC_DPP(2) QPX_FMAC_DPP(3) QPX_FMAC_DPP(4) QPX_FMAC_DPP(5)
            QPX_FMAC_DPP(6) QPX_FMAC_DPP(7) QPX_FMAC_DPP(8) QPX_FMAC_DPP(9) QPX_FMAC_DPP(10) QPX_FMAC_DPP(11)
            QPX_FMAC_DPP(12) QPX_FMAC_DPP(13) QPX_FMAC_DPP(14) QPX_FMAC_DPP(15)
#undef QPX_FMAC_DPP
            first = false;
        }
    }

    // the value lane (GK, c) holds, in every lane (g, c), g = 0 .. 3 (rows of 16 lanes).  Default: ds_bpermute (the LDS
    // crossbar, no LDS memory).  -DQPX_GRP_BCAST_SWAPS: gfx950's two lane-swap instructions (the round-2 default):  v_permlane32_swap(x, y) exchanges rows 2, 3 of x with rows 0, 1 of y; v_permlane16_swap(x, y)
    // exchanges the odd rows of x with the even rows of y -- applied to two copies of the value they leave one
    // register with the lower (even) rows everywhere and one with the upper (odd) rows.  8 instructions per
    // double, no LDS.
    template <int GK> QPX_DEV double grp_bcast(double v) const
    {
        static_assert(GK >= 0 && GK < 4, "four rows of 16 lanes");
#ifndef QPX_GRP_BCAST_SWAPS          // ds_bpermute: two instructions; with the chain wave alone on its SIMD -1 % loop time (profiles/archive/r03r)
        return __shfl(v, GK * 16 + (lane() & 15), kWave);
#else
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const long long b = __double_as_longlong(v);
        unsigned w[2] = {(unsigned)(b & 0xffffffffll), (unsigned)(b >> 32)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u2 s32 = __builtin_amdgcn_permlane32_swap(w[h], w[h], false, false);
            const unsigned x = GK < 2 ? s32.x : s32.y;
            const u2 s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
            w[h] = (GK & 1) ? s16.y : s16.x;
        }
        return __longlong_as_double(((long long)w[1] << 32) | w[0]);
#endif
    }

    // true in every lane iff `v` is true in some lane of the wave
    QPX_DEV bool any(bool v) const { return __builtin_amdgcn_ballot_w64(v) != 0ull; }

    // Issue priority of this wave among the waves of its SIMD (0 .. 3).  An f64 MFMA holds the SIMD's f64 pipe for
    // its whole duration (~83 cycles), and a wave that streams them leaves the other wave of the SIMD one vector
    // instruction per MFMA (scripts/ubench_pivot.py: a serial chain runs 15 x slower beside such a stream).  The
    // kernels therefore run their MFMA streams at priority 0 and everything else at 3: a chain instruction that is
    // ready goes before the next matrix instruction of the neighbour.
    template <int P> QPX_DEV void prio() const
    {
#ifndef QPX_NO_PRIO
        __builtin_amdgcn_s_setprio(P);
#endif
    }

    // c += A B on the matrix core, A 16x4, B 4x16, one wave (v_mfma_f64_16x16x4_f64).  Lane l gives
    // a = A[l & 15][l >> 4], b = B[l >> 4][l & 15] and holds c[r] = C[(l >> 4) + 4 r][l & 15].
    QPX_DEV void mfma16x16x4(double a, double b, double (&c)[4]) const
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc = {c[0], c[1], c[2], c[3]};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        c[0] = acc[0]; c[1] = acc[1]; c[2] = acc[2]; c[3] = acc[3];
    }
    // c += FOUR independent 4x4x4 products (v_mfma_f64_4x4x4_4b_f64), one accumulator register.  Decoded on the part
    // (scripts/probe_mfma4.py, profiles/archive/r03y_mfma4_layout.txt): lane l = 16 h + 4 blk + j holds C_blk[h][j] and gives
    // a = A_blk[j][h] (row j, k = h) and b = B_blk[h][j] (k = h, column j).  With the SAME A in all four blocks this is
    // rows 4 q .. 4 q + 3 of a 16 x 16 tile in the accumulator layout of the 16x16x4 form (register q, lane 16 h + c),
    // the B operand being register s of the other tile as it stands: the 16x16x4 product at four-row granularity
    // (16.3 cycles per instruction, profiles/archive/r03y_mfma_ubench.txt).
    QPX_DEV void mfma4x4x4(double a, double b, double& c) const { c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
    // the f32 form (v_mfma_f32_16x16x4_f32): same operands, but the accumulator layout differs -- lane
    // l holds c[r] = C[4 (l >> 4) + r][l & 15]
    QPX_DEV void mfma16x16x4(float a, float b, float (&c)[4]) const
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc = {c[0], c[1], c[2], c[3]};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        c[0] = acc[0]; c[1] = acc[1]; c[2] = acc[2]; c[3] = acc[3];
    }
    // row of the 16x16 accumulator tile that register r of lane group g holds
    static QPX_HD int mfma_row(double, int g, int r) { return g + 4 * r; }
    static QPX_HD int mfma_row(float, int g, int r) { return 4 * g + r; }
};

QPX_DEV void atomic_add_(float* p, float v) { atomicAdd(p, v); }
QPX_DEV void atomic_add_(double* p, double v) { atomicAdd(p, v); }

// Rows of 64 consecutive elements of a wave-uniform global array, read as one coalesced
// 64-lane load each: row(r) = base[r*64 + lane].  Implemented with buffer loads (SGPR resource
// + SGPR row offset + one VGPR lane offset) so that a kernel which reads ~100 different rows in
// a loop does not keep ~100 64-bit VGPR addresses alive (measured: global_load with per-row
// VGPR pairs cost 160+ VGPRs at NB = 13).
template <class T> struct GlobalRows {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;
    QPX_DEV GlobalRows(const T* base, int nelem, int lane)
    {
        // readfirstlane: make the descriptor provably wave-uniform (no waterfall loop per load)
        const unsigned long long p = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(p & 0xffffffffull));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        void* up = (void*)(((unsigned long long)hi << 32) | lo);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(up, 0, nelem * (int)sizeof(T), 0x00020000);
        voff = lane * (int)sizeof(T);
    }
    QPX_DEV T row(int r) const;
};
template <> QPX_DEV float GlobalRows<float>::row(int r) const
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, r * kWave * 4, 0));
}
template <> QPX_DEV double GlobalRows<double>::row(int r) const
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, r * kWave * 8, 0));
}

// Elements of a wave-uniform global array addressed by (per-lane element offset) + (wave-uniform element offset) + (compile-
// time element offset), widened to double: buffer loads, so that many loads share ONE 32-bit lane-offset register and the
// rest of the address lives in scalar registers / the instruction (a kernel that gathers ~40 elements per lane with
// global_load keeps ~40 64-bit addresses alive).  An offset at or beyond `nelem` reads as ZERO (the range check of a raw
// buffer: byte offset + size > num_records), which the callers use for rows of padding; offsets are never negative.
template <class S> struct GlobalBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    QPX_DEV GlobalBuf(const S* base, long long nelem)
    {
        const unsigned long long p = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(p & 0xffffffffull));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        void* up = (void*)(((unsigned long long)hi << 32) | lo);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(up, 0, (int)(nelem * (long long)sizeof(S)), 0x00020000);
    }
    QPX_DEV double at(int off) const;
    QPX_DEV void at2(int off, double (&v)[2]) const;      // elements off, off + 1: one access
};
template <> QPX_DEV double GlobalBuf<float>::at(int off) const
{
    return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off * 4, 0, 0));
}
template <> QPX_DEV double GlobalBuf<double>::at(int off) const
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off * 8, 0, 0));
}
template <> QPX_DEV void GlobalBuf<float>::at2(int off, double (&v)[2]) const
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 x = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off * 4, 0, 0));
    v[0] = (double)x[0]; v[1] = (double)x[1];
}
template <> QPX_DEV void GlobalBuf<double>::at2(int off, double (&v)[2]) const
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 x = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off * 8, 0, 0));
    v[0] = x[0]; v[1] = x[1];
}

// four consecutive elements at a 4-element-aligned address: one 128-bit access (float) or two (double)
template <class T> QPX_DEV void ld4(const T* p, T (&v)[4])
{
    const T* q = (const T*)__builtin_assume_aligned(p, 4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T));
    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
}
template <class T> QPX_DEV void st4(T* p, const T (&v)[4])
{
    T* q = (T*)__builtin_assume_aligned(p, 4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T));
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
}

template <class T> QPX_DEV T fma_(T a, T b, T c);
template <> QPX_DEV float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> QPX_DEV double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 1/x to full precision from the hardware estimate + Newton steps (the IEEE division sequence
// of `1.0/x` is ~3x longer and sits on the critical chain of every LDL column)
QPX_DEV float rcp_(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
    return r;
}
QPX_DEV double rcp_(double x)
{
#ifdef QPX_EXACT_RCP
    return 1.0 / x;
#endif
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#ifndef QPX_RCP_ONE_NEWTON
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
#endif
    return r;
}
QPX_DEV float sqrt_(float x) { return __builtin_sqrtf(x); }
QPX_DEV double sqrt_(double x) { return __builtin_sqrt(x); }
QPX_DEV float abs_(float x) { return __builtin_fabsf(x); }
QPX_DEV double abs_(double x) { return __builtin_fabs(x); }
QPX_DEV bool finite_(float x) { return __builtin_isfinite(x); }
QPX_DEV bool finite_(double x) { return __builtin_isfinite(x); }

}  // namespace qpx
#endif  // QPX_PLATFORM_H
