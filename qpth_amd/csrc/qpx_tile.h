// qpx_tile.h -- the PDIPM loop's matrix operations on MATRIX-CORE TILES (f64, gfx950).
//
// The m x m work matrix T = R + diag(s/z) lives in registers as 16x16 tiles of the lower block
// triangle, each tile in the C/D layout of v_mfma_f64_16x16x4_f64: lane l = 16 g + c of the
// owning wave holds, in register r, element (g + 4 r, c) of the tile.  Two facts about that
// layout carry the design:
//
//   * register r of a tile is a B operand (4 x 16, lane (g, c) gives B[g][c]) for the k-slice of rows g + 4 r as it
//     stands, and -- scaled per row -- an A operand (16 x 4, lane (g, c) gives A[c][g]) of the transposed tile;
//   * stored to LDS register by register ([r][lane]) a tile is plain row-major.
//
// ldl_inv (T = L~ D L~^T, with W~ = L~^-1 built in place of the eliminated columns, as in qpx_grid.h) is therefore
// BLOCKED BY SIXTEEN COLUMNS -- one tile column per panel, two barriers per panel; the scheme is described at
// panel16() below.
//
// NW waves share a QP (NW = 1, 2 or 4).  Two forms:
//   * every wave owns tile rows (CH = false): the wave that owns the pivot block eliminates it while the others wait;
//   * CHAIN-WAVE form (CH = true, round 3): one wave of the four owns no tiles.  It eliminates the pivot block of
//     panel k+1 WHILE the three tile-owning waves stream panel k's trailing updates (look-ahead by one panel: the
//     16 x 16 pivot block is a serial chain of ~3000 cycles, the trailing update of a panel ~2500 cycles of matrix
//     instructions per wave, and in the first form they ran one after the other), and it does the O(m) vector work
//     of the interior-point loop.  (An f64 MFMA holds its SIMD's vector ALU for the whole instruction: a serial chain
//     beside a matrix stream of its OWN workgroup would crawl.  The four waves of a workgroup sit on four different
//     SIMDs, so it never does; what it shares a SIMD with is a wave of the CU's other workgroup.)
// Tile rows are dealt round-robin from the bottom over the NWM tile-owning waves: wave w owns rows
// I_p = NBL-1 - p NWM - (w or NWM-1-w, alternating), p = 0 .. NPOS-1 ("positions"), and keeps tile (I_p, J) in
// register slot slot(p, J) -- a static index for static (p, J); which row a position is, is a
// wave-uniform scalar (a compile-time constant when NW = 1).  The tile row Ip of the current
// panel is a run-time value (the panel code exists once, not NBL times), tests against it are scalar branches.
//
// The triangular mat-vecs of the solve (x = -W~^T D^-1 W~ r) and the symmetric mat-vec R z use
// vector FMAs on the same registers; sums along tile rows and down tile columns go through LDS
// partials that are added in a fixed order (deterministic results).
#pragma once
#include <type_traits>

#include "qpx_grid.h"

// Sub-phase timers of one panel (-DQPX_PANEL_PROF, scripts/prof_panel.py): thread 0 of every QP adds the
// shader-clock cycles of publish / barrier / pivot block / operands / update to qpx_panel_prof[].
#ifdef QPX_PANEL_PROF
static __device__ unsigned long long qpx_panel_prof[8];   // one copy per translation unit; TU 9 reads its own
#define QPX_PP(i)                                                        \
    {                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      \
        const long long qpx_pp_n = clock64();                            \
        pacc[i] += qpx_pp_n - pacc[7];                                   \
        pacc[7] = qpx_pp_n;                                              \
    }
// chain-wave form: per wave (0 = chain wave, 1 .. 3 = tile waves) the cycles of interval 1 / wait at barrier Y /
// interval 2 / wait at barrier X, summed over panels: qpx_chain_prof[4 * wave + i]; [16] = factorisations
static __device__ unsigned long long qpx_chain_prof[20];
#define QPX_CP(i)                                                        \
    {                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      \
        const long long qpx_cp_n = clock64();                            \
        cacc[i] += qpx_cp_n - cacc[4];                                   \
        cacc[4] = qpx_cp_n;                                              \
    }
#else
#define QPX_PP(i)
#define QPX_CP(i)
#endif

namespace qpx {


// row stride of the panel's X rows (17 mod 32 doubles) and size of the region the mat-vec partials share with the
// operand tiles of the factorisation (nwm = the waves that own tiles)
QPX_LAYOUT_HD constexpr int tile_xs(int nbl) { return ((16 * nbl - 17 + 31) / 32) * 32 + 17; }
QPX_LAYOUT_HD constexpr int tile_union(int nbl, int nwm)
{
    const int npos = (nbl + nwm - 1) / nwm, a = nwm * nbl * 64 + nwm * 16 * npos * 17, b = nbl * 256;
    return a > b ? a : b;
}
QPX_LAYOUT_HD constexpr int tile_union_chain(int nbl, int nwm)      // chain-wave form: BT and its scaled copy AT
{
    return tile_union(nbl, nwm) > 2 * nbl * 256 ? tile_union(nbl, nwm) : 2 * nbl * 256;
}
// chain-wave form, behind the row buffer: S2 (256: the diagonal tile of the next pivot block but one)
constexpr int kChainExtra = 256;
// (r6) ONE wave per QP (not the chain-wave form): the operand tiles of a panel replace the panel's old rows in X where they
// stand, and everything the mat-vecs need between two factorisations lies on top of what a factorisation needs -- the
// row-sum scratch and the row sums on X, the column partials on the one buffer the pivot block goes in by and comes out of
// (S and W: one wave reads the block before it writes the factor).  With X's rows 9 mod 32 apart and the pivot block's 17
// that is 11.7 KB of scratch instead of 26 at four tile rows: with the loop's vectors 20.4 KB per QP, EIGHT workgroups on
// a CU (two to a SIMD; LDS is dealt in two halves of 80 KB, so it is 4, 6 or 8) instead of four.  TileMat::kInPlace.
QPX_LAYOUT_HD constexpr bool tile_in_place(int nwm, bool chain) { return !chain && nwm == 1; }
QPX_LAYOUT_HD constexpr int tile_xs_in_place(int nbl) { return ((16 * nbl - 9 + 31) / 32) * 32 + 9; }
QPX_LAYOUT_HD constexpr int tile_x_end_in_place(int nbl)       // X = 16 rows; { red | yrow } on top of it
{
    const int x = 16 * tile_xs_in_place(nbl), r = 16 * nbl * 17 + 16 * nbl;
    return x > r ? x : r;
}
QPX_LAYOUT_HD constexpr int tile_sw_in_place(int nbl) { return 16 * 17 > nbl * 64 ? 16 * 17 : nbl * 64; }     // S = W; part on top
QPX_LAYOUT_HD constexpr size_t tile_scratch_elems(int nbl, int nwm, bool chain = false)
{
    if (tile_in_place(nwm, chain)) return (size_t)tile_x_end_in_place(nbl) + tile_sw_in_place(nbl) + 2;
    return (size_t)16 * tile_xs(nbl) + 2 * 16 * 18 + 2 + (chain ? tile_union_chain(nbl, nwm) : tile_union(nbl, nwm)) +
           16 * (size_t)nbl + (chain ? kChainExtra : 0);
}

template <int NBL, int NW, bool CH = false> struct TileMat {
    using T = double;
    static_assert(!CH || NW == 4 || NW == 8, "the chain-wave form is one chain wave + three (loop, backward) or seven (tile sweep) tile waves");
    static constexpr int NWM = CH ? NW - 1 : NW;                           // waves that own tiles
    static constexpr int NPOS = (NBL + NWM - 1) / NWM, NT = 64 * NW, MP = 16 * NBL;
    static constexpr int psize(int p) { return NBL - p * NWM; }           // tiles of position p at most (over the waves)
    static constexpr int rowof(int p, int w) { return NBL - 1 - p * NWM - ((p & 1) ? NWM - 1 - w : w); }
    // Positions 2k and 2k+1 share one run of slots: 2k fills it from the bottom (slot = base + J), 2k+1
    // from the top (base + size - 1 - J).  In snake order their tile counts add up to the same number
    // for every wave, so nothing is wasted (NBL = 7, four tile waves: 7 slots per wave instead of 7 + 3; three: 10).
    static constexpr int pairsize(int k)
    {
        int best = 0;
        for (int w = 0; w < NWM; ++w) {
            const int a = rowof(2 * k, w) + 1 > 0 ? rowof(2 * k, w) + 1 : 0;
            const int c = (2 * k + 1 < NPOS && rowof(2 * k + 1, w) + 1 > 0) ? rowof(2 * k + 1, w) + 1 : 0;
            best = a + c > best ? a + c : best;
        }
        return best;
    }
    static constexpr int pairbase(int k)
    {
        int b = 0;
        for (int i = 0; i < k; ++i) b += pairsize(i);
        return b;
    }
    static constexpr int slot(int p, int J)
    {
        return (p & 1) ? pairbase(p / 2) + pairsize(p / 2) - 1 - J : pairbase(p / 2) + J;
    }
    static constexpr int NSLOT = pairbase((NPOS + 1) / 2);
    static constexpr int NROW = 16 * NPOS;                                 // matrix rows a wave owns (at most)
    struct Pos {
        int tid, lane, w, g, c;      // w: index among the tile-owning waves (-1: the chain wave)
        bool chain;
        QPX_DEV explicit Pos(const Block& blk)
            : tid(blk.tid), lane(blk.lane()), w(NW == 1 ? 0 : blk.uniform(blk.wave()) - (CH ? 1 : 0)), g(blk.lane() >> 4),
              c(blk.lane() & 15), chain(CH && blk.uniform(blk.wave()) == 0)
        {
        }
        // Chain-wave form: wave 0 is the chain wave.  (Round 3 first assigned the roles by the SIMD a wave runs on
        // -- s_getreg HW_ID -- so that the chain waves of the two workgroups of a CU shared SIMD 0 and never sat beside
        // the other workgroup's MFMA streams; by wave index they land on different SIMDs, each beside one tile wave of
        // the other QP.  Same box, C2: 0.5349 vs 0.5327 ms -- no difference, so the simpler rule stays:
        // profiles/archive/r03a (placement probe), r03q (A/B).)
        QPX_DEV void assign(const Block&, int*) {}
        // the wave that does the O(m) vector work of the interior-point loop
        QPX_DEV bool lead(const Block& blk) const { return CH ? chain : blk.wave() == 0; }
        QPX_DEV bool is_chain() const { return CH && chain; }
        QPX_DEV int wi() const { return w; }                 // index among the tile waves
        // the same position with lane coordinates the optimiser cannot trace back (see QPX_LAUNDER_V)
        QPX_DEV Pos fresh() const
        {
            Pos q = *this;
            QPX_LAUNDER_V(q.lane);
            QPX_LAUNDER_V(q.g);
            QPX_LAUNDER_V(q.c);
            return q;
        }
        // tile row of position p (< 0: none).  Rows are dealt from the bottom in snake order (w, then
        // NWM-1-w, ...) so that the tile counts of the waves stay close as the factorisation retires rows
        QPX_DEV int row(int p) const
        {
            if (CH && chain) return -1;
            return NBL - 1 - p * NWM - ((p & 1) ? NWM - 1 - w : w);      // = rowof(p, w)
        }
    };
    // A position whose ROLE is a compile-time constant: -1 = the chain wave, 0 .. NWM-1 = that tile wave -- row(p) is
    // then a constant and every "does this wave own ..." test in the functions below folds away.  The kernels of the
    // chain-wave form run their whole body once per role (with_role): with run-time roles the four waves' different
    // tile sets meet in the same registers behind scalar branches, which cost the factorisation forty 64-bit moves per
    // panel at the loop's back edge and kept the MFMA chains of different tiles from being interleaved.
    template <int ROLE> struct RolePos : Pos {
        QPX_DEV explicit RolePos(const Pos& q) : Pos(q) {}
        QPX_DEV constexpr int row(int p) const { return ROLE < 0 ? -1 : rowof(p, ROLE < 0 ? 0 : ROLE); }
        QPX_DEV constexpr bool lead(const Block&) const { return ROLE < 0; }
        QPX_DEV constexpr bool is_chain() const { return ROLE < 0; }
        QPX_DEV constexpr int wi() const { return ROLE; }
        QPX_DEV RolePos fresh() const { return RolePos(Pos::fresh()); }
    };
    template <class P> struct role_of { static constexpr int value = -2; };                    // run-time role
    template <int ROLE> struct role_of<RolePos<ROLE>> { static constexpr int value = ROLE; };
    // f(position): once with the run-time position, or (chain-wave form) with this wave's role as a constant
    template <int R, class F> static QPX_DEV void with_tile_role(const Pos& p, F&& f)
    {
        if constexpr (R + 1 < NWM) {
            if (p.w == R) f(RolePos<R>(p));
            else with_tile_role<R + 1>(p, f);
        } else {
            f(RolePos<R>(p));
        }
    }
    // INVARIANT of the role forms: f is instantiated once per role behind wave-divergent branches, so every role
    // executes its OWN copy of each s_barrier -- legal on this hardware because s_barrier counts arrivals of the
    // workgroup's waves whatever their program counters, and correct only as long as EVERY role issues the identical
    // sequence of workgroup barriers.  A barrier under an `is_chain()` / `row()` test breaks it.  The host-thread
    // emulator checks exactly this: its barrier is a counting one per workgroup, and a barrier that not every thread
    // reaches is a reported dead-lock (tests/emu/qpx_emu.cpp: fiber_deadlock), not a hang; the ThreadSanitizer build
    // runs every role on its own pthreads.
    template <class F> static QPX_DEV void with_role(const Pos& p, F&& f)
    {
        if constexpr (CH) {
            if (p.chain) f(RolePos<-1>(p));
            else with_tile_role<0>(p, f);
        } else {
            f(p);
        }
    }
    struct Regs { T e[NSLOT][4]; };
    // (r6) Chain-wave form of the LOOP kernel: the chain wave runs ahead of the tile waves across the loop's phases too.
    // d = s/z of a pass is final when the previous pass ends, and T(0,0) = R(0,0) + diag(d) is R's tile (constant, kept in
    // the chain wave's registers: it owns no tiles) plus d on the diagonal.  So while the tile waves load R and form R z', the chain wave eliminates pivot block 0; the residual /
    // best-iterate / stop bookkeeping of the pass (which needs R z') runs on it in the second interval of panels 0 and 1,
    // where the tile waves' nine-tile updates are the longer side and the chain wave used to wait at the barrier.  Until
    // round 5 these ran one after the other: mat-vec | vector work | pivot block 0 (three tile waves waiting through the
    // last two).  The chain wave is the critical resource of a pass: what counts is that nothing is ADDED to its serial
    // work where it is the longer side (a first version that also moved its share of panel 0's first interval in front of
    // the mat-vec's barrier made that phase chain-bound: 6.2 k cycles against the tile waves' 4.5 k, and bought 2 % instead
    // of 8: profiles/r06c_phases.txt).  QPX_NO_AHEAD restores the round-5 order (A/B).
#ifdef QPX_NO_AHEAD
    static constexpr bool kAhead = false;
#else
    static constexpr bool kAhead = CH;
#endif
#ifndef QPX_PIVOT_HEAD
#define QPX_PIVOT_HEAD 2
#endif
    // pivots of the next block the chain wave eliminates ahead of barrier Y (same box, C2 loop kernel: 0 -> 0.4749 ms,
    // 2 -> 0.4709, 4 -> 0.4863, 6 -> 0.5010, 8 -> 0.5128: profiles/archive/r03vb -- the wait it fills is two pivots long)
    static constexpr int kPivotHead = QPX_PIVOT_HEAD;
    // scratch: X (16 x XS: the 16 old rows of a panel) | S, W (16 x SS: pivot block, its inverse factor) | flag |
    // { part (NWM x NBL x 64) | red (NWM x NROW x 17) } or, during a factorisation, BT (NBL x 256: the operand
    // tiles; chain-wave form: + AT, the same tiles times -1/d) | yrow (MP) | chain-wave form: S2 (256: the diagonal
    // tile after next).  XS = 17 mod 32 and SS = 18 keep both the row-wise and the transposed accesses
    // (lane stride XS resp. SS doubles) on distinct LDS banks.
    // kInPlace (one wave per QP): operand tile b_J is written over X_J -- same shape, same (row, column) addressing, the
    // wave that reads X_J is the one that writes b_J --; S and W are one buffer; { red | yrow } lie on top of X and the
    // column partials on top of S / W (they live between factorisations, X, S and W inside one): no BT, no mat-vec scratch
    // of its own (tile_scratch_elems).
    static constexpr bool kInPlace = tile_in_place(NWM, CH);
    static constexpr int XS = kInPlace ? tile_xs_in_place(NBL) : tile_xs(NBL), SS = kInPlace ? 17 : 18;
    static constexpr int kXEnd = kInPlace ? tile_x_end_in_place(NBL) : 16 * XS;
    static constexpr int kX = 0, kS = kXEnd, kW = kInPlace ? kS : kS + 16 * SS;
    static constexpr int kFlag = kInPlace ? kS + tile_sw_in_place(NBL) : kW + 16 * SS;
    static constexpr int kPart = kInPlace ? kS : kFlag + 2;
    static constexpr int kRed = kInPlace ? 0 : kPart + NWM * NBL * 64, kBT = kPart, kAT = kBT + NBL * 256;
    static constexpr int kRow = kInPlace ? NWM * NROW * 17 : kPart + (CH ? tile_union_chain(NBL, NWM) : tile_union(NBL, NWM));
    static constexpr int kS2 = kRow + MP;
    static_assert(!kInPlace || (kRow + MP <= kXEnd && NWM * NBL * 64 <= tile_sw_in_place(NBL) && NROW == 16 * NBL), "one-wave layout");
    // operand tile J, register r of lane p: where it is kept between the operand phase and the update
    static QPX_DEV int bt_at(const Pos& p, int J, int r)
    {
        return kInPlace ? kX + (p.g + 4 * r) * XS + 16 * J + p.c : kBT + J * 256 + r * 64 + p.lane;
    }
    QPX_LAYOUT_HD static size_t scratch_elems() { return kInPlace ? (size_t)kFlag + 2 : (size_t)kRow + MP + (CH ? kChainExtra : 0); }
    static QPX_DEV void sync(const Block& blk)
    {
        if (NW == 1) blk.wave_sync();
        else blk.sync();
    }
    static QPX_DEV const T* image(const T* F, const FacLayout& lay) { return F + lay.Rm; }

    // img: tile (I, J), J <= I, at [(I (I + 1) / 2 + J) * 256 + r * 64 + lane]
    template <class P> static QPX_DEV void load(const Block& blk, const P& p0, Regs& E, const T* img)
    {
        const P p = p0.fresh();
        const GlobalRows<T> rows(img, NBL * (NBL + 1) / 2 * 256, p.lane);
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J <= I) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) E.e[slot(pp, J)][r] = rows.row((I * (I + 1) / 2 + J) * 4 + r);
                }
            }
        }
    }

    template <class P> static QPX_DEV void add_diag(const P& p, Regs& E, const T* vd)
    {
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J != I) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (p.c == p.g + 4 * r) E.e[slot(pp, J)][r] += vd[16 * J + p.c];
            }
        }
    }

    // Sums over the 16 columns of every tile row of this wave: acc[pp][r] of lane (g, c) is a partial of
    // matrix row 16 I_pp + g + 4 r.  Through LDS: one padded line of 17 per row, one lane adds a line.
    template <class P, class F>
    static QPX_DEV void row_reduce(const Block& blk, const P& p, const T (&acc)[NPOS][4], T* scr, F&& emit, int red_off = kRed)
    {
        T* red = scr + red_off + p.wi() * (NROW * 17);
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((pp * 4 + r) * 4 + p.g) * 17 + p.c] = acc[pp][r];
        blk.wave_sync();
#pragma unroll
        for (int o0 = 0; o0 < NROW; o0 += 64) {
            const int o = o0 + p.lane;                       // o = (pp * 4 + r) * 4 + g
            if (o < NROW) {
                const T* q = red + o * 17;
                const T s = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]))) +
                            (((q[8] + q[9]) + (q[10] + q[11])) + ((q[12] + q[13]) + (q[14] + q[15])));
                const int I = p.row(o >> 4);
                if (I >= 0) emit(16 * I + (o & 3) + 4 * ((o >> 2) & 3), s);
            }
        }
    }

    // out[j] = +-(base[j] + the column partials of every tile wave), fixed order
    template <bool kNeg>
    static QPX_DEV void gather_cols(const Block& blk, const T* part, const T* base, T* out)
    {
        for (int j = blk.tid; j < MP; j += NT) {
            const int J = j >> 4, cc = j & 15;
            T sum = base[j];
#pragma unroll
            for (int w = 0; w < NWM; ++w) {
                const T* pp = part + (size_t)(w * NBL + J) * 64 + cc;
                sum += (pp[0] + pp[16]) + (pp[32] + pp[48]);
            }
            out[j] = kNeg ? -sum : sum;
        }
    }

    // the same by one wave (the chain wave, when it is the only consumer of the result: no barrier behind it)
    template <bool kNeg>
    static QPX_DEV void gather_cols_wave(const Block& blk, int lane, const T* part, const T* base, T* out)
    {
        for (int j = lane; j < MP; j += kWave) {
            const int J = j >> 4, cc = j & 15;
            T sum = base[j];
#pragma unroll
            for (int w = 0; w < NWM; ++w) {
                const T* pp = part + (size_t)(w * NBL + J) * 64 + cc;
                sum += (pp[0] + pp[16]) + (pp[32] + pp[48]);
            }
            out[j] = kNeg ? -sum : sum;
        }
        blk.wave_sync();
    }

    // vout = S vin for the symmetric matrix in E (diagonal tiles hold both triangles).  kLeadOnly: only the lead wave
    // reads vout, and not before the workgroup's next barrier is anything in scr written again (chain-wave form: the
    // chain wave gathers by itself, one barrier less).
    template <bool kLeadOnly = false, class P>
    static QPX_DEV void symv(const Block& blk, const P& p0, const Regs& E, const T* vin, T* vout, T* scr)
    {
        const P p = p0.fresh();
        T* part = scr + kPart;
        T* yrow = scr + kRow;
        if (!p.is_chain()) {
            T acc[NPOS][4], u[NPOS][4];
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[pp][r] = T(0);
                    u[pp][r] = I >= 0 ? vin[16 * I + p.g + 4 * r] : T(0);
                }
            }
#pragma unroll
            for (int J = 0; J < NBL; ++J) {
                const T xj = vin[16 * J + p.c];
                T col = 0;
#pragma unroll
                for (int pp = 0; pp < NPOS; ++pp) {
                    if (J >= psize(pp)) continue;
                    const int I = p.row(pp);
                    if (J > I) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[pp][r] = fma_(E.e[slot(pp, J)][r], xj, acc[pp][r]);
                    if (J < I) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) col = fma_(E.e[slot(pp, J)][r], u[pp][r], col);
                    }
                }
                part[(size_t)(p.wi() * NBL + J) * 64 + p.lane] = col;
            }
            row_reduce(blk, p, acc, scr, [&](int i, T s) { yrow[i] = s; });
        }
        sync(blk);
        if constexpr (kLeadOnly && CH && role_of<P>::value >= -1) {
            if constexpr (role_of<P>::value == -1) gather_cols_wave<false>(blk, p.lane, part, yrow, vout);
        } else {
            gather_cols<false>(blk, part, yrow, vout);
            sync(blk);
        }
    }

    // ---- ldl_inv BLOCKED BY SIXTEEN COLUMNS (one tile column per panel, two barriers per panel).
    //
    // Panel Ip = rows/columns 16 Ip .. 16 Ip + 15, pivot block P = E(Ip, Ip) = L~_pp D L~_pp^T, W_pp = L~_pp^-1:
    //   publish   X_J (16 x 16, J = 0 .. NBL-1, J != Ip) = the panel's sixteen "old" rows: the W~ entries E(Ip, J) left
    //             of the panel (from the wave that owns tile row Ip), the panel's columns read down the matrix,
    //             E(J, Ip)^T, right of it (from the owners of those tiles; the update restarts them from zero).
    //   factor    one wave moves P to "lane (g, c) = row c, columns 4 g .. 4 g + 3" (through LDS) and eliminates it
    //             there: per pivot one v_rcp_f64_dpp + Newton, the multipliers copied to
    //             the four lane groups by lane swaps, and ONE v_fmac_f64_dpp per register -- the pivot row arrives
    //             through the DPP row broadcast; nothing is published, no barrier.  The same rank-1 update builds W~
    //             in place of the eliminated columns (as everywhere in this file).
    //   operands  b_J = W_pp X_J on the matrix core (4 MFMAs per J, the J dealt over the waves; b_Ip = W_pp), written
    //             to LDS in the accumulator layout.
    //   update    E(Ip, J) = b_J (the panel's own rows are final);  E(I, J) += (-D^-1 b_I)^T b_J for I > Ip, J <= I:
    //             register r of b_J is the B operand of k-slice r as it is, and register r of b_I times -1/d is the
    //             A operand (the accumulator layout indexes both by (row g + 4 r, column c)).
    // Order in time, CH = false (panel16):  publish | factor (the owner of tile row Ip; the others wait) -- barrier A --
    // operands -- barrier B -- update.  CH = true (ldl_inv_chain): see there.
    //
    // Pivot K of the 16 x 16 pivot block held as lane (g, c) = row c, columns 4 g .. 4 g + 3 (four registers) plus,
    // in every group, the row's own diagonal entry dg (updated by dg -= l~^2 d, which needs nothing from other
    // lanes): the pivot d_K is then lane K's dg in every group, so its reciprocal (the long chain: estimate + two
    // Newton steps) runs beside the lane swaps that copy column K from its group K / 4 to the other three.  Every
    // lane then updates its four columns with the pivot row from lane K of its own row of 16 lanes.
    // (r3) The copy of column K to the other three lane groups used to sit ON the chain of the sixteen pivots (rank-1
    // update of pivot K-1 -> select -> ds_bpermute, ~80 cycles -> multipliers -> rank-1 update of pivot K).  Now every
    // group carries the NEXT pivot column `vn` itself: the copy of column K+1 as it stands is started at the top of
    // pivot K, and pivot K's update is applied to the copy by one more multiply-add through the row broadcast -- the same
    // operation, on the same values, as the owner's -- so that only the reciprocal chain links one pivot to the next.
    // kMixed: a block may hold pivots of both signs (the pre-factorisation with equality constraints), and which KIND broke down
    // is reported: the reciprocal is then captured at its pivot (after a breakdown NaNs flow back into the diagonal entries
    // of earlier pivots); everywhere else the sixteen reciprocals are taken once, behind the block (pivot_store).
    template <int K, bool kMixed = false>
    static QPX_DEV void pivot16(const Block& blk, const Pos& p, T (&a)[4], T& dg, T& myr, T& vn)
    {
        constexpr int GK = K / 4, KK = K % 4;
        const T dk = blk.template row_bcast<K>(dg);
        T raw[1] = {T(0)};
        if constexpr (K < 14) raw[0] = blk.template grp_bcast<(K + 1) / 4>(a[(K + 1) % 4]);   // column K+1 before this pivot's update
        const T r = rcp_(dk);                                    // (the sixteen reciprocals are stored and checked together, after the block: pivot_store)
        if constexpr (kMixed) myr = p.lane == K ? r : myr;
        if constexpr (K < 15) {
            const T v = p.c > K ? vn : T(0);                     // column K below the pivot, 0 above
            const T nl = -(v * r);                               // -l~
            blk.template row_rank1<K>(a, nl);
            a[KK] = p.g == GK ? nl : a[KK];                      // column K: assigned (see qpx_grid.h); 0 on and above the diagonal
            dg = fma_(nl, v, dg);                                // (moved in front of the rank-1 update: +1 % loop time, r03h)
            if constexpr (K < 14) {
                blk.template row_rank1<K>(raw, nl);              // raw[c] += raw[K] * nl[c]: column K+1 after this pivot
                vn = raw[0];
            }
        } else {
            a[KK] = p.g == GK ? T(0) : a[KK];
        }
    }

    // The pivot block in S (LDS, row-major, stride SS) -> its unit-lower inverse factor in W (strictly lower part), the
    // reciprocals of its pivots in rd[k0 ..], flag[0] = 1 if a pivot is not positive and finite.  One wave; kmax = the
    // pivots that are not identity padding (>= 16: all; the padded ones are skipped: d = 1, no multipliers, their
    // columns of W are zero).
    // sign: +1 / -1 = the pivots of this block must all be positive / negative, 2 + k = the first k positive and the
    // rest negative (the equality rows of the pre-factorisation, qpx_prefac.h); flag[0] = 1 / 2 when a pivot that must be
    // positive / negative is not.
    // (in three steps, so that the chain wave can eliminate the first kPivotHead pivots of the NEXT block in the interval
    // of a panel in which it otherwise waits for the tile waves' operand tiles -- factor_role)
    struct PivotState { T a[4]; T dg, myr, vn; };
    static QPX_DEV void pivot_load(const Block& blk, const Pos& p, const T* scr, PivotState& st)
    {
        const T* S = scr + kS;
#pragma unroll
        for (int j = 0; j < 4; ++j) st.a[j] = S[p.c * SS + 4 * p.g + j];
        st.dg = S[p.c * SS + p.c];
        st.myr = T(1);
        st.vn = blk.template grp_bcast<0>(st.a[0]);         // column 0, in every lane group
    }
    template <int K0, int K1, bool kMixed = false>
    static QPX_DEV void pivot_run(const Block& blk, const Pos& p, PivotState& st, int kmax)
    {
        static_for<K1 - K0>([&](auto kc) {
            constexpr int K = K0 + decltype(kc)::value;
            if (K == 0 || kmax > K) pivot16<K, kMixed>(blk, p, st.a, st.dg, st.myr, st.vn);
        });
    }
    template <bool kMixed = false>
    static QPX_DEV void pivot_store(const Block& blk, const Pos& p, T* scr, T* rd, int k0, int kmax, int sign, const PivotState& st0)
    {
        T* W = scr + kW;
        T* flag = scr + kFlag;
        // (r6) lane c's dg is final once pivot c has been taken (later pivots add nl * v with v = 0 there): the sixteen
        // reciprocals at once, from the very values the pivots broadcast -- until round 6 every pivot selected its reciprocal
        // into lane K (a compare and two selects per pivot on the chain); bit-identical results, C3's loop -3 %, C5's -2 %
        // (profiles/r06y_ab_pivot_reciprocals_at_the_end.txt).  Skipped (padded) pivots keep 1.
        PivotState st = st0;
        if constexpr (!kMixed) st.myr = (p.lane < 16 && p.lane < kmax) ? rcp_(st0.dg) : T(1);        // (lanes 0 .. 15 hold and check them, as before)
#pragma unroll
        for (int j = 0; j < 4; ++j) W[p.c * SS + 4 * p.g + j] = (4 * p.g + j < kmax) ? st.a[j] : T(0);
        if (p.lane < 16) rd[k0 + p.lane] = st.myr;
        // a pivot that is not positive and finite leaves a reciprocal that is not (negative, NaN from inf - inf or
        // 0 * inf further down, 0 or inf); nothing above traps, so one test of the sixteen reciprocals replaces
        // two compares in every pivot's chain
        // sign >= 2 (round 4, the pre-factorisation with equality constraints): a block whose first sign - 2 pivots
        // must be positive and the rest negative
        const int split = sign >= 2 ? sign - 2 : (sign > 0 ? 16 : 0);
        const bool neg = p.lane < 16 && p.lane < kmax && p.lane >= split;
        const T sr = neg ? -st.myr : st.myr;                               // (skipped pivots keep myr = 1)
        const bool wrong = !(sr > T(0) && sr < T(1e300));
        const bool badp = blk.any(wrong && !neg), badn = blk.any(wrong && neg);
        if (p.lane == 0) flag[0] = badp ? T(1) : (badn ? T(2) : T(0));
    }
    template <bool kMixed = false>
    static QPX_DEV void pivot_block(const Block& blk, const Pos& p, T* scr, T* rd, int k0, int kmax, int sign = 1)
    {
        PivotState st;
        pivot_load(blk, p, scr, st);
        pivot_run<0, 16, kMixed>(blk, p, st, kmax);
        pivot_store<kMixed>(blk, p, scr, rd, k0, kmax, sign, st);
    }

    // ---- the chain wave ahead of the tile waves across the loop's phases (kAhead; see there)
    struct Ahead {
        T a00[4], d00;       // R(0,0) in the pivot layout (lane (g, c): row c, columns 4 g .. 4 g + 3) and its diagonal entry
    };
    template <class P> static QPX_DEV void ahead_init(const Block&, const P& p0, Ahead& ah, const T* img)
    {
        if constexpr (role_of<P>::value == -1) {
            const P p = p0.fresh();
            // element (i, j) of tile t sits at img[256 t + (i >> 2) 64 + 16 (i & 3) + j]; diagonal tiles hold both triangles
#pragma unroll
            for (int j = 0; j < 4; ++j) ah.a00[j] = img[p.g * 64 + 16 * j + p.c];                          // (4 g + j, c) = (c, 4 g + j)
            ah.d00 = img[(p.c >> 2) * 64 + 16 * (p.c & 3) + p.c];
        }
    }
    // chain wave, while the tile waves load R and form R z': pivot block 0 of T = R + diag(vd) -> W, rd[0..15], flag
    template <class P> static QPX_DEV void ahead_pivot0(const Block& blk, const P& p0, const Ahead& ah, const T* vd, T* scr, T* rd, int m)
    {
        if constexpr (role_of<P>::value == -1) {
            const Pos p = p0.fresh();
            PivotState st;
            const T dc = vd[p.c];
#pragma unroll
            for (int j = 0; j < 4; ++j) st.a[j] = ah.a00[j] + ((4 * p.g + j == p.c) ? dc : T(0));
            st.dg = ah.d00 + dc;
            st.myr = T(1);
            st.vn = blk.template grp_bcast<0>(st.a[0]);
            pivot_run<0, 16>(blk, p, st, m);
            pivot_store(blk, p, scr, rd, 0, m, 1, st);
        }
    }
    // tile waves, in front of the factorisation's first barrier: the partial sums of R vin (symv's, but with the column
    // partials BEHIND the row-sum scratch: the chain wave adds them up in the factorisation's first interval, while the
    // tile waves already write the operand tiles BT -- which share LDS with the usual place of the partials); then
    // T = R + diag(s/z) -- d from z and s themselves, the very operations that give vD its values: nothing of this
    // depends on the chain wave, so no barrier separates it from the mat-vec -- and the sixteen old rows of panel 0 -> X
    static constexpr int kPartA = kPart + NWM * NROW * 17;
    static_assert(!CH || (kPartA >= kBT + NBL * 256 && kPartA + NWM * NBL * 64 <= kRow), "kAhead: the column partials must survive the first operand tiles");
    template <class P> static QPX_DEV void ahead_front(const Block& blk, const P& p0, Regs& E, const T* vin, const T* vz, const T* vs, T* scr)
    {
        if constexpr (role_of<P>::value >= 0) {
            const P p = p0.fresh();
            T* part = scr + kPartA;
            T* yrow = scr + kRow;
            T acc[NPOS][4], u[NPOS][4];
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[pp][r] = T(0);
                    u[pp][r] = I >= 0 ? vin[16 * I + p.g + 4 * r] : T(0);
                }
            }
#pragma unroll
            for (int J = 0; J < NBL; ++J) {
                const T xj = vin[16 * J + p.c];
                T col = 0;
#pragma unroll
                for (int pp = 0; pp < NPOS; ++pp) {
                    if (J >= psize(pp)) continue;
                    const int I = p.row(pp);
                    if (J > I) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[pp][r] = fma_(E.e[slot(pp, J)][r], xj, acc[pp][r]);
                    if (J < I) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) col = fma_(E.e[slot(pp, J)][r], u[pp][r], col);
                    }
                }
                part[(size_t)(p.wi() * NBL + J) * 64 + p.lane] = col;
            }
            row_reduce(blk, p, acc, scr, [&](int i, T s) { yrow[i] = s; }, kPart);
            // T = R + diag(s / z)
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int J = 0; J < psize(pp); ++J) {
                    if (J != I) continue;
                    const T d = vs[16 * J + p.c] * rcp_(vz[16 * J + p.c]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (p.c == p.g + 4 * r) E.e[slot(pp, J)][r] += d;
                }
            }
            publish_rows<role_of<P>::value>(p0.fresh(), E, scr, 0, false);
        }
    }
    // the same without a mat-vec (the KKT kernels: factor_kkt for a given d): T = R + diag(vd), panel 0's old rows -> X
    template <class P> static QPX_DEV void ahead_publish0(const Block&, const P& p0, Regs& E, const T* vd, T* scr)
    {
        if constexpr (role_of<P>::value >= 0) {
            add_diag(p0, E, vd);
            publish_rows<role_of<P>::value>(p0.fresh(), E, scr, 0, false);
        }
    }
    // chain wave, behind that barrier: vout = R vin from the partials
    static QPX_DEV void ahead_gather(const Block& blk, int lane, const T* scr, T* vout)
    {
        gather_cols_wave<false>(blk, lane, scr + kPartA, scr + kRow, vout);
    }

    // The panel's sixteen old rows -> X (and, with_s, the pivot block itself -> S), from the tiles this wave owns.
    // Chain-wave form: also the NEXT diagonal tile, E(Ip + 1, Ip + 1) as it stands -> S2 (accumulator layout).
    static QPX_DEV void publish(const Pos& p, const Regs& E, T* scr, int Ip, bool with_s, bool& mine)
    {
        T* X = scr + kX;
        T* S = scr + kS;
        if constexpr (CH) {
            // picked by value, stored once: the same store under ten branches is merged by the compiler into one store
            // through a pointer into the tile array, and a tile array whose address is taken lives in scratch memory
            T* S2 = scr + kS2;
            T d2[4] = {T(0), T(0), T(0), T(0)};
            bool have = false;
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int J = 0; J < psize(pp); ++J) {
                    const bool hit = I == Ip + 1 && J == Ip + 1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2[r] = hit ? E.e[slot(pp, J)][r] : d2[r];
                    have = have || hit;
                }
            }
            if (have) {
#pragma unroll
                for (int r = 0; r < 4; ++r) S2[r * 64 + p.lane] = d2[r];
            }
        }
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            const int I = p.row(pp);
            if (I < Ip) continue;
            if (I == Ip) mine = true;
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J > Ip) continue;
                if (I == Ip) {
                    if (J < Ip) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[(p.g + 4 * r) * XS + 16 * J + p.c] = E.e[slot(pp, J)][r];
                    } else {
                        if (with_s) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) S[(p.g + 4 * r) * SS + p.c] = E.e[slot(pp, J)][r];
                        }
                        if constexpr (CH) {      // the panel's own block of X: the identity, so that b_Ip = W_pp X_Ip like every other b_J
#pragma unroll
                            for (int r = 0; r < 4; ++r) X[(p.g + 4 * r) * XS + 16 * J + p.c] = (p.g + 4 * r == p.c) ? T(1) : T(0);
                        }
                    }
                } else if (J == Ip) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[p.c * XS + 16 * I + p.g + 4 * r] = E.e[slot(pp, J)][r];
                }
            }
        }
    }

    // Operand tile b_J = (I + W_strict) X_J of panel Ip (b_Ip = I + W_strict itself) -> acc; wa = the A operand of W
    static QPX_DEV void operand_tile(const Block& blk, const Pos& p, const T* scr, int Ip, int J, const T (&wa)[4], T (&acc)[4])
    {
        const T* X = scr + kX;
        const T* W = scr + kW;
        if (J == Ip) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = W[(p.g + 4 * r) * SS + p.c] + ((p.g + 4 * r == p.c) ? T(1) : T(0));
        } else {
            T bx[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = bx[s] = X[(p.g + 4 * s) * XS + 16 * J + p.c];
#pragma unroll
            for (int s = 0; s < 4; ++s) blk.mfma16x16x4(wa[s], bx[s], acc);
        }
    }

    // Update phase of panel Ip from the operand tiles in BT: the panel's own rows are final (read straight into the
    // tile registers: an assignment from registers another path also uses costs round-trip copies of the tile), every
    // owned tile (I, J), I > Ip, J <= I gets its rank-16 update -- except tile (skip, skip) (chain-wave form: the next
    // pivot block, which the chain wave brings up to date on its own copy).  zr: a zero the compiler cannot see
    // through.  Chain-wave form: the A operands come scaled from AT (a wave holds ten tiles there; three tile rows of
    // operands in registers on top of them is what sent tiles to scratch memory).
    static QPX_DEV void update(const Block& blk, const Pos& p, Regs& E, const T* scr, const T* rd, int Ip, int skip, T zr)
    {
        const T* BT = scr + kBT;
        const T* AT = scr + kAT;
        const int k0 = 16 * Ip;
        T aop[CH ? 1 : NPOS][4];
        if constexpr (!CH) {
            T nrd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) nrd[r] = -rd[k0 + p.g + 4 * r];
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int r = 0; r < 4; ++r) aop[pp][r] = T(0);
                if (I <= Ip) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) aop[pp][r] = scr[bt_at(p, I, r)] * nrd[r];
            }
        }
#pragma unroll
        for (int pp = 0; pp < NPOS; ++pp) {
            if (p.row(pp) != Ip) continue;
#pragma unroll
            for (int J = 0; J < psize(pp); ++J) {
                if (J > Ip) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) E.e[slot(pp, J)][r] = scr[bt_at(p, J, r)];
            }
        }
#pragma unroll
        for (int J = 0; J < NBL; ++J) {
            bool need = false;
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                if (J >= psize(pp)) continue;
                const int I = p.row(pp);
                need = need || (I > Ip && J <= I && !(I == skip && J == skip));
            }
            if (!need) continue;
            T bj[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bj[r] = scr[bt_at(p, J, r)];
            // the panel's own columns restart from zero (their old values went into X): multiplied by a factor that is
            // 0 there and 1 elsewhere, outside every branch
            const T keep = J == Ip ? zr : T(1);
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                if (J >= psize(pp)) continue;
                const int I = p.row(pp);
                if (I > Ip && J <= I && !(I == skip && J == skip)) {
                    if constexpr (CH) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) aop[0][r] = AT[I * 256 + r * 64 + p.lane];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) E.e[slot(pp, J)][r] *= keep;
#pragma unroll
                    for (int s = 0; s < 4; ++s) blk.mfma16x16x4(aop[CH ? 0 : pp][s], bj[s], E.e[slot(pp, J)]);
                }
            }
        }
    }

    // Chain-wave form, rows of tile wave W as compile-time constants: the sixteen old rows of panel Ip -> X (with_s: the
    // pivot block -> S), the identity into the panel's own block of X, and the next diagonal tile -> S2
    template <int W, int PP>
    static QPX_DEV void publish_row(const Pos& p, const Regs& E, T* scr, int Ip, bool with_s)
    {
        constexpr int I = rowof(PP, W);
        if constexpr (I >= 0) {
            T* X = scr + kX;
            T* S = scr + kS;
            T* S2 = scr + kS2;
            if (I == Ip + 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) S2[r * 64 + p.lane] = E.e[slot(PP, I)][r];
            }
            if (I == Ip) {
#pragma unroll
                for (int J = 0; J < I; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[(p.g + 4 * r) * XS + 16 * J + p.c] = E.e[slot(PP, J)][r];
                if (with_s) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[(p.g + 4 * r) * SS + p.c] = E.e[slot(PP, I)][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(p.g + 4 * r) * XS + 16 * I + p.c] = (p.g + 4 * r == p.c) ? T(1) : T(0);
            } else if (I > Ip) {
                // tile (I, Ip), transposed; Ip is a run-time column of this row: picked by value (see publish)
                T t[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
                for (int J = 0; J < I; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[r] = J == Ip ? E.e[slot(PP, J)][r] : t[r];
#pragma unroll
                for (int r = 0; r < 4; ++r) X[p.c * XS + 16 * I + p.g + 4 * r] = t[r];
            }
        }
    }
    template <int W>
    static QPX_DEV void publish_rows(const Pos& p, const Regs& E, T* scr, int Ip, bool with_s)
    {
        publish_row<W, 0>(p, E, scr, Ip, with_s);
        if constexpr (NPOS > 1) publish_row<W, 1>(p, E, scr, Ip, with_s);
        if constexpr (NPOS > 2) publish_row<W, 2>(p, E, scr, Ip, with_s);
    }

    // Chain-wave form: the update phase with the wave's tile rows as COMPILE-TIME constants (W = index of the tile wave;
    // the caller switches on it).  The run-time form above puts every tile behind scalar branches of its own: four
    // dependent MFMAs (95 cycles each) behind a fresh LDS load, ~570 cycles per tile.  Here a tile row is one
    // straight-line block -- operands loaded up front, the MFMAs of its tiles interleaved (k-slice outermost), 83
    // cycles per MFMA -- and only the row as a whole, the restart of the panel's own column and the tile the chain wave
    // has taken over sit behind (scalar) conditions.
    // kSweep (the round-3 pre-factorisation on matrix-core tiles, deleted in round 4 after it lost its A/B; the parameter
    // stays false in every instantiation -- the branches below are kept because removing them from the headline kernel's
    // source for tidiness is not worth a regression): the symmetric sweep instead of the elimination -- EVERY tile row gets the
    // update (rows above the panel accumulate the negated inverse of the swept block), the panel's own row becomes
    // (-D^-1 b_Ip)^T b_J instead of b_J.
    template <int W, int PP, bool kSweep>
    static QPX_DEV void update_row(const Block& blk, const Pos& p, Regs& E, const T* scr, int Ip, int skip, T zr, int mrows, const T (&nrd)[4])
    {
        constexpr int I = rowof(PP, W);
        if constexpr (I >= 0) {
            const T* BT = scr + kBT;
            const T* AT = scr + kAT;
            if constexpr (I >= NBL - 2 && NBL >= 4 && !kSweep) {      // (the chain-wave form serves 65 <= m <= 112: tile rows 5 and 6 can be part padding)
                // The last tile rows hold the matrix's last rows and then padding (m = 100: four real rows of sixteen in
                // tile row 6), and they are the widest: seven and six tiles, updated by every panel.  When it is part padding it is updated at
                // four-row granularity with the four-block matrix instruction (v_mfma_f64_4x4x4_4b: register q of a tile
                // += A_q B with the same B operands), and row blocks that are all padding get nothing -- their operand
                // columns are zero.  (Everywhere else the 16x16x4 form stays: with full tiles the four-block form
                // measured 6 % slower, profiles/archive/r03x.)
                const int nq = mrows - 16 * I >= 16 ? 4 : (mrows - 16 * I <= 0 ? 0 : (mrows - 16 * I + 3) >> 2);
                if (I > Ip && nq < 4) {
                    if (nq > 0) update_row_quads<PP, I>(blk, p, E, BT, nrd, Ip, skip, zr, nq);
                    return;
                }
            }
            if (I == Ip && !kSweep) {            // the panel's own rows are final
#pragma unroll
                for (int J = 0; J <= I; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) E.e[slot(PP, J)][r] = BT[J * 256 + r * 64 + p.lane];
            } else if (I > Ip || kSweep) {
                T a[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = kSweep ? AT[I * 256 + r * 64 + p.lane] : nrd[r] * BT[I * 256 + r * 64 + p.lane];
                // off-diagonal tiles, in groups of at most four (registers), then the diagonal one
                constexpr int G = NSLOT > 12 ? 2 : 4;          // (interleaved MFMA chains per group: registers)
#pragma unroll
                for (int J0 = 0; J0 < I; J0 += G) {
                    constexpr int dummy = 0;
                    T b[G][4];
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const int J = J0 + j;
                        if (J >= I) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) b[j][r] = BT[J * 256 + r * 64 + p.lane];
                        // the panel's own column (and, sweep, its own row) restarts from zero
                        const T keep = (J == Ip || (kSweep && I == Ip)) ? zr : T(1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) E.e[slot(PP, J)][r] *= keep;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const int J = J0 + j;
                            if (J >= I) continue;
                            blk.mfma16x16x4(a[s], b[j][s], E.e[slot(PP, J)]);
                        }
                }
                if (I != skip) {
                    T b[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[r] = BT[I * 256 + r * 64 + p.lane];
                    if (kSweep) {
                        const T keep = I == Ip ? zr : T(1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) E.e[slot(PP, I)][r] *= keep;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) blk.mfma16x16x4(a[s], b[s], E.e[slot(PP, I)]);
                }
            }
        }
    }
    // tile row I (> Ip), row blocks 0 .. nq - 1 of every tile: a[q][s] = the scaled operand tile's entries
    // (4 s + h, 4 q + j) in every block of four lanes of lane row h -- the A operand of the four-block instruction
    template <int PP, int I>
    static QPX_DEV void update_row_quads(const Block& blk, const Pos& p, Regs& E, const T* BT, const T (&nrd)[4], int Ip, int skip, T zr, int nq)
    {
        const T* ATq = BT + I * 256 + (p.lane & ~15) + (p.lane & 3);      // (scaled by -1/d below: row 4 s + h of the operand tile, h = this lane's row)
        constexpr int G = 4;                         // tiles whose products are interleaved
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= nq) break;                      // (uniform)
            T a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = nrd[s] * ATq[s * 64 + 4 * q];
#pragma unroll
            for (int J0 = 0; J0 <= I; J0 += G) {
                T b[G][4];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int J = J0 + j;
                    if (J > I) continue;
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[j][s] = BT[J * 256 + s * 64 + p.lane];
                    if (J < I) E.e[slot(PP, J)][q] *= (J == Ip) ? zr : T(1);      // the panel's own column restarts from zero
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const int J = J0 + j;
                        if (J > I) continue;
                        if (J == I && I == skip) continue;       // (the chain wave brings the next pivot block up to date itself)
                        blk.mfma4x4x4(a[s], b[j][s], E.e[slot(PP, J)][q]);
                    }
            }
        }
    }
    template <int W, bool kSweep = false>
    static QPX_DEV void update_rows(const Block& blk, const Pos& p, Regs& E, const T* scr, int Ip, int skip, T zr, int mrows, const T (&nrd)[4])
    {
        // heaviest row last: its tiles are the ones the publish that follows does not read
        if constexpr (NPOS > 2) update_row<W, 2, kSweep>(blk, p, E, scr, Ip, skip, zr, mrows, nrd);
        if constexpr (NPOS > 1) update_row<W, 1, kSweep>(blk, p, E, scr, Ip, skip, zr, mrows, nrd);
        update_row<W, 0, kSweep>(blk, p, E, scr, Ip, skip, zr, mrows, nrd);
    }

    // Chain-wave form: two operand tiles b_J = X_J + W_strict X_J at once (J0, J1 run-time; J1 < 0: one), their MFMA
    // chains interleaved; -> BT and (kScaled: the tile sweep's updates read it), times -1/d, -> AT.  The factorisation's
    // updates scale their A operands themselves (update_row): sixteen LDS writes less per tile wave in the interval
    // of a panel that the tile waves bound, four multiplications more in the one the pivot block bounds.
    // kSlices: only the first ns k-slices of four pivots are not padding (the last panel of a matrix whose order is not a
    // multiple of sixteen: W_pp is the identity there and the panel's old rows are zero)
    template <bool kScaled = true, bool kSlices = false>
    static QPX_DEV void operand_pair(const Block& blk, const Pos& p, T* scr, int J0, int J1, const T (&wa)[4], const T (&nrd)[4], int ns = 4)
    {
        const T* X = scr + kX;
        T* BT = scr + kBT;
        T* AT = scr + kAT;
        const int Jb = J1 < 0 ? J0 : J1;
        T x0[4], x1[4], c0[4], c1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0[s] = x0[s] = X[(p.g + 4 * s) * XS + 16 * J0 + p.c];
            c1[s] = x1[s] = X[(p.g + 4 * s) * XS + 16 * Jb + p.c];
        }
        QPX_SCHED_FENCE();                      // (the scheduler otherwise runs the two chains one after the other)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (kSlices && s >= ns) break;          // (uniform)
            blk.mfma16x16x4(wa[s], x0[s], c0);
            blk.mfma16x16x4(wa[s], x1[s], c1);
            QPX_SCHED_FENCE();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            BT[J0 * 256 + r * 64 + p.lane] = c0[r];
            if constexpr (kScaled) AT[J0 * 256 + r * 64 + p.lane] = nrd[r] * c0[r];
        }
        if (J1 >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                BT[J1 * 256 + r * 64 + p.lane] = c1[r];
                if constexpr (kScaled) AT[J1 * 256 + r * 64 + p.lane] = nrd[r] * c1[r];
            }
        }
    }

    static QPX_DEV bool panel16(const Block& blk, const Pos& p, Regs& E, T* scr, T* rd, int Ip, int m, long long (&pacc)[8])
    {
        QPX_PP(5)
        T* W = scr + kW;
        T* BT = scr + kBT;
        T* flag = scr + kFlag;
        const int k0 = 16 * Ip;
        const int kmax = m - k0;              // pivots of this block that are not identity padding (>= 16: all)
        bool mine = false;
        publish(p, E, scr, Ip, true, mine);
        QPX_PP(0)
        // -- the pivot block, by the wave that owns it
        if (mine) {
            blk.wave_sync();
            pivot_block(blk, p, scr, rd, k0, kmax);
        }
        QPX_PP(1)
        sync(blk);
        QPX_PP(2)
        const T zr = flag[0];                 // 0 from here on
        if (zr != T(0)) return false;
        blk.template prio<0>();               // the matrix-instruction streams yield to the other wave's chains
        // -- operand tiles: b_J = (I + W_strict) X_J
        {
            T wa[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wa[s] = W[p.c * SS + p.g + 4 * s];
#pragma unroll
            for (int jj = 0; jj < (NBL + NW - 1) / NW; ++jj) {
                const int J = p.w + jj * NW;
                if (J >= NBL) continue;
                T acc[4];
                operand_tile(blk, p, scr, Ip, J, wa, acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[bt_at(p, J, r)] = acc[r];
            }
        }
        QPX_PP(3)
        sync(blk);
        QPX_PP(4)
        update(blk, p, E, scr, rd, Ip, -1, zr);
        blk.template prio<3>();
        return true;
    }

    // ---- CHAIN-WAVE FORM.  Look-ahead by one panel; per panel k two barriers, X and Y:
    //
    //             chain wave                                    tile waves (two operand tiles each)
    //   (after X) b_{k+1} = W_k X_{k+1} -> BT, AT;               b_J = W_k X_J -> BT, -D^-1 b_J -> AT   (J != k+1)
    //             S = S2 + (-D^-1 b_{k+1})^T b_{k+1}: pivot
    //             block k+1, up to date
    //   -- barrier Y --
    //             PIVOT BLOCK k+1 (S -> W, rd, flag)             the updates of panel k (all but tile (k+1, k+1), whose
    //                                                            only reader was the pivot block); panel k's own rows; the
    //                                                            sixteen old rows of panel k+1 -> X, E(k+2, k+2) -> S2
    //   -- barrier X --
    //
    // The serial chain of a factorisation is then  pivot block -> b_{k+1} -> one tile update -> pivot block  inside ONE
    // wave (~4000 cycles per panel on an idle CU), with the ~9 tile updates per tile wave and panel (~3000 cycles of
    // MFMAs) beside it instead of behind it.  The chain wave needs from the tile waves only two tiles per panel --
    // E(k+1, k)^T (part of X) and E(k+1, k+1) (S2) as they stand after panel k-1 -- published one panel ahead.
    // LDS: every buffer is written in one interval and read in the next, never both in one.
    // ROLE: -1 = the chain wave, 0 .. NWM-1 = tile wave (its rows are compile-time constants in here: the whole
    // factorisation exists once per role, selected once -- with the roles told apart by scalar branches inside the
    // panel loop the register allocator no longer kept the tiles in place across the back edge: forty 64-bit moves per
    // panel and wave).
    // `panel(k)` = (pivots of panel k that are not identity padding, +1 / -1: their sign); npan panels.  kSweep: the
    // symmetric sweep of the first npan tile rows (update_row).  Returns 0, or the flag of the pivot block that failed.
    template <int ROLE, bool kSweep, bool kMixed = false, class PanelInfo>
    static QPX_DEV int factor_role(const Block& blk, const Pos& p0, Regs& E, T* scr, T* rd, int npan, int mrows, PanelInfo&& panel)
    {
        return factor_role_impl<ROLE, kSweep, false, kMixed>(blk, p0, E, scr, rd, npan, mrows, panel, [] {}, [] {}, [] {}, [] { return 0; });
    }
    // kA (the loop kernel, kAhead): pivot block 0 is done (ahead_pivot0), X and S2 hold panel 0's old rows and E(1, 1)
    // (ahead_front) and a barrier lies behind both.  The chain wave runs `extra0` (it adds up the mat-vec's partial sums) at
    // the top of panel 0's first interval, `extra1` (the loop's residual / best-iterate / stop bookkeeping) behind pivot
    // block 1 in panel 0's second interval and `extra2` (the affine right-hand side) in panel 1's; behind panel 0's barrier X every wave asks `stopped()` and the flag of pivot block 0: -1 = the loop stops
    // (the panel of speculative work is dropped), > 0 = pivot block 0 broke down.
    template <int ROLE, bool kSweep, bool kA, bool kMixed = false, class PanelInfo, class Extra0, class Extra1, class Extra2, class Stopped>
    static QPX_DEV int factor_role_impl(const Block& blk, const Pos& p0, Regs& E, T* scr, T* rd, int npan, int mrows, PanelInfo&& panel,
                                        Extra0&& extra0, Extra1&& extra1, Extra2&& extra2, Stopped&& stopped)
    {
        constexpr bool kChain = ROLE < 0;
        constexpr int W = kChain ? 0 : ROLE;
        QPX_LAUNDER_S(npan);
        Pos p = p0.fresh();
        T* S = scr + kS;
        T* W_ = scr + kW;
        T* BT = scr + kBT;
        T* AT = scr + kAT;
        T* S2 = scr + kS2;
        T* flag = scr + kFlag;
        // -- panel 0: its rows and pivot block -> X, S (and E(1, 1) -> S2); then the pivot block
        if constexpr (!kA) {
            if constexpr (!kChain) publish_rows<W>(p, E, scr, 0, true);
            blk.sync();
            if constexpr (kChain) pivot_block<kMixed>(blk, p, scr, rd, 0, panel(0).kmax, panel(0).sign);
            blk.sync();
        }
        long long cacc[5] = {0, 0, 0, 0, 0};
#ifdef QPX_PANEL_PROF
        cacc[4] = clock64();
#endif
        PivotState pst;                       // (chain wave: the next pivot block, carried across the barrier between the intervals)
#pragma unroll 1
        for (int k = 0; k < npan; ++k) {
            const T zr = flag[0];             // 0 unless a pivot broke down
            if (!(kA && k == 0) && zr != T(0)) return (int)zr;      // (kA: pivot block 0's flag is looked at behind barrier X, after `extra1`)
            const bool la = k + 1 < npan;     // there is a next pivot block
            p = p0.fresh();
            // ---- interval 1: operand tiles; the chain wave brings the next pivot block up to date
            {
                T wa[4], nrd[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) wa[s] = W_[p.c * SS + p.g + 4 * s];
#pragma unroll
                for (int r = 0; r < 4; ++r) nrd[r] = -rd[16 * k + p.g + 4 * r];
                if constexpr (kChain) {
                    if constexpr (kA) {
                        if (k == 0) extra0();
                    }
                    if (la) {
                        // (two accumulators per product -- chains of two dependent MFMAs instead of four -- measured no
                        // gain: this interval waits for the tile waves' operand tiles anyway, profiles/archive/r03h)
                        const T* X = scr + kX;
                        T acc[4], bx[4], ao[4], sacc[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            acc[s] = bx[s] = X[(p.g + 4 * s) * XS + 16 * (k + 1) + p.c];
                            sacc[s] = S2[s * 64 + p.lane];
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s) blk.mfma16x16x4(wa[s], bx[s], acc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) ao[r] = nrd[r] * acc[r];
#pragma unroll
                        for (int s = 0; s < 4; ++s) blk.mfma16x16x4(ao[s], acc[s], sacc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            S[(p.g + 4 * r) * SS + p.c] = sacc[r];
                            BT[(k + 1) * 256 + r * 64 + p.lane] = acc[r];
                            if constexpr (kSweep) AT[(k + 1) * 256 + r * 64 + p.lane] = ao[r];
                        }
                        // ... and starts on it: the tile waves' operand tiles take longer than this wave's two products,
                        // the first pivots fill the wait (W, rd and the flag are written after the barrier: the tile
                        // waves still read this panel's)
                        blk.wave_sync();
                        pivot_load(blk, p, scr, pst);
                        pivot_run<0, kPivotHead, kMixed>(blk, p, pst, panel(k + 1).kmax);
                    }
                } else {
                    // the tile waves share the other operand tiles, two each: entries W and W + NWM of the list of the
                    // J != k + 1 (the last panel has no k + 1: one more entry, for tile wave 0)
                    blk.template prio<0>();
                    constexpr int e0 = W, e1 = W + NWM;
                    const int J0 = (la && e0 > k) ? e0 + 1 : e0, J1 = (la && e1 > k) ? e1 + 1 : e1;
                    if (la || kSweep) {
                        operand_pair<kSweep>(blk, p, scr, J0, J1 < NBL ? J1 : -1, wa, nrd);
                    } else {                     // the last panel: its padded pivots' k-slices contribute nothing
                        const int km = panel(k).kmax, ns = km >= 16 ? 4 : (km + 3) >> 2;
                        operand_pair<kSweep, true>(blk, p, scr, J0, J1 < NBL ? J1 : -1, wa, nrd, ns);
                    }
                    // (entries beyond the first two per wave: NBL > 2 NWM, or the last panel's one extra entry)
#pragma unroll
                    for (int e2 = W + 2 * NWM; e2 < NBL; e2 += NWM) {
                        const int J2 = (la && e2 > k) ? e2 + 1 : e2;
                        if (J2 < NBL) operand_pair<kSweep>(blk, p, scr, J2, -1, wa, nrd);
                    }
                    blk.template prio<3>();
                }
            }
            QPX_CP(0)
            blk.sync();
            QPX_CP(1)
            p = p0.fresh();
            // ---- interval 2: the chain wave eliminates pivot block k+1, the tile waves stream panel k's updates
            if constexpr (kChain) {
                if (la) {
                    pivot_run<kPivotHead, 16, kMixed>(blk, p, pst, panel(k + 1).kmax);
                    pivot_store<kMixed>(blk, p, scr, rd, 16 * (k + 1), panel(k + 1).kmax, panel(k + 1).sign, pst);
                }
                if constexpr (kA) {
                    if (k == 0) extra1();
                    if (k == 1 || (k == 0 && npan < 2)) extra2();
                }
            } else {
                blk.template prio<0>();
                T nrd[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) nrd[r] = -rd[16 * k + p.g + 4 * r];
                update_rows<W, kSweep>(blk, p, E, scr, k, la ? k + 1 : -1, zr, mrows, nrd);
                if (la) publish_rows<W>(p, E, scr, k + 1, false);
                blk.template prio<3>();
            }
            QPX_CP(2)
            blk.sync();
            QPX_CP(3)
            if (kA && k == 0) {
                if (stopped()) return -1;
                if (zr != T(0)) return (int)zr;
            }
        }
#ifdef QPX_PANEL_PROF
        if (p0.lane == 0) {
            const int wv = kChain ? 0 : 1 + W;
            if (wv < 4)
                for (int i = 0; i < 4; ++i) atomicAdd(&qpx_chain_prof[4 * wv + i], (unsigned long long)cacc[i]);
            if (kChain) atomicAdd(&qpx_chain_prof[16], 1ull);
        }
#endif
        return (int)flag[0];
    }
    // (A form of the chain-wave factorisation WITHOUT workgroup barriers between chain and tile waves -- LDS words
    // signalling "W_k is ready" one way and "the inputs of pivot block k are ready" the other, the tile waves keeping a
    // counter barrier of their own, so that the chain wave never waits for the operand tiles -- was built and measured
    // in round 3: +7 % loop time (0.560 vs 0.523 ms at C2, profiles/archive/r03s_ab_async_chain.txt).  The tile waves, not the
    // chain wave, are the longer side of a panel (operand tiles + nine tile updates + publication ~ 4 700 cycles against
    // ~4 100 for pivot block + look-ahead), so taking the chain wave off their barriers buys nothing and the polling
    // costs a little.  Removed.)
    struct PanelOf { int kmax, sign; };
    template <int ROLE>
    static QPX_DEV bool ldl_inv_role(const Block& blk, const Pos& p0, Regs& E, T* scr, T* rd, int m)
    {
        const int npan = (m + 15) / 16 < NBL ? (m + 15) / 16 : NBL;
        return factor_role<ROLE, false>(blk, p0, E, scr, rd, npan, m, [m](int k) { return PanelOf{m - 16 * k, 1}; }) == 0;
    }

    // the loop kernel's factorisation with the chain wave ahead (kAhead): 0 = done, -1 = the loop stops, > 0 = breakdown
    template <class P, class Extra0, class Extra1, class Extra2, class Stopped>
    static QPX_DEV int ldl_inv_ahead(const Block& blk, const P& p, Regs& E, T* scr, T* rd, int m, Extra0&& extra0, Extra1&& extra1, Extra2&& extra2,
                                     Stopped&& stopped)
    {
        static_assert(CH && role_of<P>::value >= -1, "chain-wave form: call through with_role");
        blk.template prio<3>();
        const int npan = (m + 15) / 16 < NBL ? (m + 15) / 16 : NBL;
        return factor_role_impl<role_of<P>::value, false, true>(blk, p, E, scr, rd, npan, m, [m](int k) { return PanelOf{m - 16 * k, 1}; },
                                                               extra0, extra1, extra2, stopped);
    }

    // E: T (SPD, order m, padded with the identity) -> strictly lower: W~ = L~^-1, rd[k] = 1/d_k; false: a pivot
    // broke down (uniform)
    template <class P> static QPX_DEV bool ldl_inv(const Block& blk, const P& p, Regs& E, T* scr, T* rd, int m)
    {
        bool ok = true;
        blk.template prio<3>();
        if constexpr (CH) {
            static_assert(role_of<P>::value >= -1, "chain-wave form: call through with_role");
            ok = ldl_inv_role<role_of<P>::value>(blk, p, E, scr, rd, m);
        } else {
            long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef QPX_PANEL_PROF
            pacc[7] = clock64();
#endif
#pragma unroll 1
            for (int Ip = 0; Ip < NBL && ok && 16 * Ip < m; ++Ip) ok = panel16(blk, p, E, scr, rd, Ip, m, pacc);
#ifdef QPX_PANEL_PROF
            if (p.tid == 0) {
                for (int i = 0; i < 6; ++i) atomicAdd(&qpx_panel_prof[i], (unsigned long long)pacc[i]);
                atomicAdd(&qpx_panel_prof[6], 1ull);
            }
#endif
            sync(blk);
        }
        return ok;
    }

    // vout = -T^-1 vin = -W~^T D^-1 W~ vin (W~ unit lower in E, strictly lower part stored)
    template <bool kLeadOnly = false, class P>
    static QPX_DEV void solve_neg(const Block& blk, const P& p0, const Regs& E, const T* rd, int m, const T* vin,
                                  T* vout, T* tmp, T* scr)
    {
        const P p = p0.fresh();
        T* part = scr + kPart;
        if (!p.is_chain()) {
            // u = D^-1 W~ vin: sums along tile rows, complete inside the owning wave
            T acc[NPOS][4];
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[pp][r] = T(0);
#pragma unroll
            for (int J = 0; J < NBL; ++J) {
                const T xj = vin[16 * J + p.c];
#pragma unroll
                for (int pp = 0; pp < NPOS; ++pp) {
                    if (J >= psize(pp)) continue;
                    const int I = p.row(pp);
                    if (J > I) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const T e = (J < I || p.c < p.g + 4 * r) ? E.e[slot(pp, J)][r] : T(0);
                        acc[pp][r] = fma_(e, xj, acc[pp][r]);
                    }
                }
            }
            row_reduce(blk, p, acc, scr, [&](int i, T s) { tmp[i] = (i < m) ? (s + vin[i]) * rd[i] : T(0); });
            blk.wave_sync();      // a wave owns whole tile rows: the u it reads next are the ones it just wrote
            // x = W~^T u: sums down columns, partial per wave, gathered in a fixed order
            T u[NPOS][4];
#pragma unroll
            for (int pp = 0; pp < NPOS; ++pp) {
                const int I = p.row(pp);
#pragma unroll
                for (int r = 0; r < 4; ++r) u[pp][r] = I >= 0 ? tmp[16 * I + p.g + 4 * r] : T(0);
            }
#pragma unroll
            for (int J = 0; J < NBL; ++J) {
                T col = 0;
#pragma unroll
                for (int pp = 0; pp < NPOS; ++pp) {
                    if (J >= psize(pp)) continue;
                    const int I = p.row(pp);
                    if (J > I) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const T e = (J < I || p.c < p.g + 4 * r) ? E.e[slot(pp, J)][r] : T(0);
                        col = fma_(e, u[pp][r], col);
                    }
                }
                part[(size_t)(p.wi() * NBL + J) * 64 + p.lane] = col;
            }
        }
        sync(blk);
        if constexpr (kLeadOnly && CH && role_of<P>::value >= -1) {
            if constexpr (role_of<P>::value == -1) gather_cols_wave<true>(blk, p.lane, part, tmp, vout);
        } else {
            gather_cols<true>(blk, part, tmp, vout);
            sync(blk);
        }
    }
};

QPX_LAYOUT_HD size_t lds_elems_ipm_tile(int nbl, int nw, int n, int q, bool chain = false)
{
    return lds_elems_ipm_loop(16 * (size_t)nbl, tile_scratch_elems(nbl, chain ? nw - 1 : nw, chain), n, q);
}

QPX_LAYOUT_HD size_t lds_elems_kkt_tile(int nbl, int nw, int n, int q, bool chain = false)
{
    return lds_elems_kkt_mat(16 * (size_t)nbl, tile_scratch_elems(nbl, chain ? nw - 1 : nw, chain), n, q);
}

template <int NBL, int NW, bool kBackward, bool CH = false>
QPX_DEV void kkt_tile_body(const Block& b, const KktArgs<double>& a, int qp, double* lds)
{
    kkt_mat_body<double, TileMat<NBL, NW, CH>, kBackward>(b, a, qp, lds);
}

template <int NBL, int NW, int NS, bool CH = false>
QPX_DEV void ipm_tile_body(const Block& b, const IpmArgs<double>& a, int qp, double* lds)
{
    ipm_loop_body<double, TileMat<NBL, NW, CH>, NS>(b, a, qp, lds);
}

}  // namespace qpx
