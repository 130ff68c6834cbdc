// qpx_launch.h -- launcher declarations shared by the kernel translation units and the C ABI.
// Each (kernel, dtype) pair is compiled in its own translation unit (qpx_hip_kernels.hip with
// -DQPX_TU_KERNEL=... -DQPX_TU_REAL=...) so the build parallelises; the definitions live there.
#pragma once
#include <cstddef>
#include <type_traits>

#include "qpx_kernels.h"
#include "qpx_grid.h"
#include "qpx_tile.h"
#include "qpx_prefac.h"
#include "qpx_reduce.h"
#include "qpx_big.h"
#include "qpx_big_polish.h"

namespace qpx {

template <int V> using Int = std::integral_constant<int, V>;
template <bool V> using Bool = std::integral_constant<bool, V>;

inline size_t lds_budget_bytes() { return kMaxLdsBytes; }

// thread-grid kernels (qpx_grid.h), 16x16 threads per QP, format-3 blob
template <class T, int NBL> int launch_sweep(const PrefactorArgs<T>& a, size_t lds_bytes, void* stream);
template <class T, int NBL, int NS> int launch_ipm_grid(const IpmArgs<T>& a, size_t lds_bytes, void* stream);
template <int NBN, bool kEq> int launch_prefac_tile(const PrefactorArgs<double>& a, size_t lds_bytes, void* stream);     // pre_factor_kkt on matrix-core tiles (qpx_prefac.h)
template <int NBL, int NW, int NS, bool CH = false> int launch_ipm_tile(const IpmArgs<double>& a, size_t lds_bytes, void* stream);     // matrix-core tiles, f64, NW waves per QP (CH: chain-wave form)
template <int NBL, int NW, bool kBw, bool CH = false> int launch_kkt_tile(const KktArgs<double>& a, size_t lds_bytes, void* stream);
template <class T, int NBL, int NS> int launch_ipm_grid8(const IpmArgs<T>& a, size_t lds_bytes, void* stream);   // 8x8 grid = one wave
template <class T, int NBL, bool kBw> int launch_kkt_grid(const KktArgs<T>& a, size_t lds_bytes, void* stream);

// the finishing stage (qpx_polish): thread grid (any dtype) and matrix-core tiles (f64)
template <class T, int NBL> int launch_polish_grid(const PolishArgs<T>& a, size_t lds_bytes, void* stream);
template <int NBL, int NW, bool CH> int launch_polish_tile(const PolishArgs<double>& a, size_t lds_bytes, void* stream);

// batch-mean outer products of shared-parameter gradients (qpx_reduce.h): one workgroup of 16 waves per output tile
template <class T> int launch_batch_outer(const OuterArgs<T>& a, int tiles, void* stream);
template <class T> int launch_dense_solve(const DenseSolveArgs<T>& a, void* stream);      // one general k x k system per workgroup (factor_solve_kkt_reg)

// the large-QP family (qpx_big.h): every launch covers the batch; gy = workgroups per QP
template <class T> int launch_big_pack(const BigPackArgs<T>& a, int gy, void* stream);
template <class T> int launch_big_panel(const BigPanelArgs<T>& a, void* stream);
template <class T> int launch_big_gemm(const BigGemmArgs<T>& a, void* stream);
template <class T> int launch_big_trsv(const BigTrsvArgs<T>& a, void* stream);
template <class T> int launch_big_gemv(const BigGemvArgs<T>& a, void* stream);
template <class T> int launch_big_symv(const BigSymvArgs<T>& a, void* stream);
template <class T> int launch_big_vec(const BigVecArgs<T>& a, void* stream);
template <class T> int launch_big_kkt(const BigKktArgs<T>& a, int gy, void* stream);
template <class T> int launch_big_phase(const BigPhaseArgs<T>& a, void* stream);
template <class T> int launch_big_solve(const BigSolveArgs<T>& a, void* stream);
template <class T> int launch_big_diag(const BigDiagArgs<T>& a, void* stream);
template <class T> int launch_big_polish(const BigPolishArgs<T>& a, void* stream);      // the finishing stage (qpx_big_polish.h)
// Side streams for the parts of a batch the large-QP family works on concurrently (qpx_api.inc: big_split).
// stream_fork: side[0 .. nside) = streams of the calling host thread's pool, made to wait (event) for everything
// enqueued on `caller` so far; stream_join: `caller` waits (events) for everything enqueued on them.  Neither
// synchronises the host -- except that stream_fork checks a side stream ONCE per caller stream (outside stream capture) for
// running beside it and not on its hardware queue (qpx_hip_api.hip: side_slot, ~0.3 ms).  delay_us > 0: side stream i first sleeps i * delay_us (one idle wave), which sets the
// parts out of phase with each other.
// `first`: pool slot of side[0] -- [0, kMaxSide) are the parts' streams, kMaxSide + i the helper stream of part i's loop.
int stream_fork(void* caller, int nside, void** side, int delay_us, int first = 0);
int stream_join(void* caller, int nside, void* const* side, int first = 0);

}  // namespace qpx
