#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic QPs that is already resident
in HBM:  z = QPFunction(verbose=-1)(Q, p, G, h, A, b); z.backward(ones)   (forward + backward,
the measurement defined in SURVEY.md section 8d / prof-linear.py:110-118, device-synchronised).
Workload = BASELINE.json configs[1] extended with the backward pass as its `metric` asks:
batch=512, nz=100, nineq=100, neq=0 per GPU, float64 (the dtype the 1e-4 parity gate holds in).
With N GPUs every rank solves its own 512-QP shard (the batch dimension shards with no
data-path collective) => weak scaling; value = N*512*K / max-over-ranks time.

The JSON line also carries
  roofline ...... the dominant kernel (k_ipm, the PDIPM loop): algorithmic bytes per launch /
                  average launch duration measured live with HIP events on the launch stream
  cpu_baseline .. the oracle (C restatement of the reference's CPU path) timed on this host
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import problems  # noqa: E402
from qpth_amd.qp import QPFunction  # noqa: E402
from qpth_amd.kkt import KKTFactors, set_stall_policy  # noqa: E402
from qpth_amd import _lib  # noqa: E402

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
# Dense f64 / f32 peaks.  f32: 157.3 TF (MI355X_MICROARCH.md, matrix = vector).  f64: the guide has no
# row for it; 78.6 TF is AMD's MI355X datasheet number for both the vector and the matrix pipe, and our
# micro-benchmarks agree: v_fma_f64 issues every 4.07 ticks per wave (77 TF over 1024 SIMDs at the
# measured 2.39 GHz tick), v_mfma_f64_16x16x4 every 83 ticks (60 TF) -- scripts/ubench_mfma.py.
MFMA_PEAK = {"f64": 78.6e12, "f32": 157.3e12}


def algorithmic_flops_per_qp(n, m, q, iters):
    """SURVEY.md section 8(d), "ALGORITHMIC flops per QP (Cholesky-minimum)", the part the loop kernel
    replaces: per IPM iteration m^3/3 (factor) + 2 x (4n^2 + 4mn + 4qn + 2m^2) (two solves) + 2n^2 + 4mn + 4qn
    (residuals); the start point is one more factorisation + solve.  (The kernel itself spends 2x the
    factorisation flops -- it builds the inverse factor in place -- and far fewer on solves and residuals,
    which it does in the m-dimensional condensed space; the roofline prices the algorithm, not the kernel.)"""
    solve = 4.0 * n * n + 4.0 * m * n + 4.0 * q * n + 2.0 * m * m
    resid = 2.0 * n * n + 4.0 * m * n + 4.0 * q * n
    fact = m ** 3 / 3.0
    return iters * (fact + 2.0 * solve + resid) + (fact + solve)


def algorithmic_bytes_per_qp(n, m, q, w):
    """compulsory HBM traffic per QP (forward read+write, backward read+write)."""
    fwd_r = w * (n * n + m * n + q * n + n + m + q)
    fwd_w = w * (n + 2 * m + q)
    bwd_r = w * (n * n + m * n + q * n + 2 * n + 2 * m + q)
    bwd_w = w * (n * n + n + m * n + m + q * n + q)
    return fwd_r, fwd_w, bwd_r, bwd_w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--nz", type=int, default=100)
    ap.add_argument("--nineq", type=int, default=100)
    ap.add_argument("--neq", type=int, default=0)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fixed-iters", action="store_true",
                    help="also time the loop kernel with early stopping off (all maxIter iterations); off by "
                         "default so that every launch of the loop kernel in a profiled run does the same work")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path to time.")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    B, n, m, q = args.batch, args.nz, args.nineq, args.neq
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    w = 8 if args.dtype == "f64" else 4
    # every rank owns its own shard: different seed per rank, same generator (prof-linear.py:64-75)
    Q, p, G, h, A, b = problems.prof_qp(B, n, m, q, seed=rank, dtype=np_dt)
    tQ, tp, tG, th, tA, tb = [torch.tensor(x, device=dev) for x in (Q, p, G, h, A, b)]
    tp.requires_grad_(True)                      # prof-linear.py:99
    ones = torch.ones(B, n, dtype=tQ.dtype, device=dev)
    qpf = QPFunction(verbose=-1)
    if os.environ.get("QPX_VARIANT"):            # A/B knob for kernel development (include/qpx.h)
        _lib.hip().dll.qpx_set_ipm_variant(int(os.environ["QPX_VARIANT"]))

    def step():
        z = qpf(tQ, tp, tG, th, tA, tb)
        z.backward(ones)
        tp.grad = None
        return z

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        z = step()
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    if rank == 0:
        # ---- per-kernel timing with HIP events on the launch stream (torch's current stream) ----
        fac = KKTFactors.build(tQ, tG, tA, B)
        res = fac.ipm(tp.detach(), th, tb)
        torch.cuda.synchronize()
        iters_mean = float(res.iters.float().mean().item())
        nrep = max(5, min(args.steps, 30))

        def time_launches(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(nrep):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / nrep * 1e-3

        t_pre = time_launches(lambda: KKTFactors.build(tQ, tG, tA, B))
        t_ipm = time_launches(lambda: fac.ipm(tp.detach(), th, tb))
        t_bwd = time_launches(lambda: fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones))
        t_ipm_fixed = None
        if args.fixed_iters:
            set_stall_policy(_lib.STALL_OFF)
            t_ipm_fixed = time_launches(lambda: fac.ipm(tp.detach(), th, tb))
            set_stall_policy(None)

        fwd_r, fwd_w, bwd_r, bwd_w = algorithmic_bytes_per_qp(n, m, q, w)
        ipm_bytes = (fwd_r + fwd_w) * B          # the forward's compulsory traffic, DESIGN.md section 6
        # The loop kernel is compute-side bound (55 flop per compulsory byte at C2, machine balance ~10):
        # its roofline is the dense matrix/vector peak of the dtype it computes in.
        ipm_flops = algorithmic_flops_per_qp(n, m, q, iters_mean) * B
        achieved = ipm_flops / t_ipm
        peak = MFMA_PEAK[args.dtype]
        traffic = None
        tf = os.path.join(ROOT, "profiles", "ipm_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if rec.get("config") == [B, n, m, q, args.dtype]:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": "PDIPM loop kernel (k_ipm_tile / k_ipm_grid, one launch per forward)", "bound": "mfma",
                    "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic, "algorithmic_flops_per_launch": ipm_flops,
                    "algorithmic_bytes_per_launch": ipm_bytes, "hbm_frac": ipm_bytes / t_ipm / HBM_PEAK,
                    "launch_ms": t_ipm * 1e3}

        cpu_baseline = None
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is reported at N = 1 only
            from oracle import qp_oracle as orc
            ncpu = os.cpu_count() or 1
            np1 = np.ones((B, n), np_dt)

            def cpu_pass(nth):
                c0 = time.perf_counter()
                o = orc.OracleQP(Q, p, G, h, A, b, nthreads=nth)
                x, y, lam, s, info = o.forward()                      # reference (batch-global) semantics
                o.backward(x, lam, s, y, np1)
                return time.perf_counter() - c0, info

            # OpenMP over QPs: more threads than physical cores (or than memory channels can feed) is
            # slower, so scan a few team sizes once and keep the best; `cores` is the size that won
            best, cores, info = None, 1, None
            for nth in sorted({c for c in (8, 16, 32, 64, 128) if c <= ncpu} | {min(ncpu, 4)}):
                t, inf = cpu_pass(nth)
                if best is None or t < best:
                    best, cores, info = t, nth, inf
            for rep in range(2):
                t, info = cpu_pass(cores)
                best = min(best, t)
            cpu_baseline = {"value": B / best, "unit": "QPs/s", "cores": cores, "kind": "port",
                            "sample": "the full workload once (B=%d fwd+bwd, best of 3 at the best OpenMP team size "
                                      "of a scan over 4..128 threads, %s, %d IPM iterations, %d host CPUs)"
                                      % (B, args.dtype, int(info["trips"]), ncpu)}

        out = {
            "metric": "QPs/sec (fwd+bwd) at batch=512 nz=100 nineq=100; 1/2/4/8 MI355X",
            "value": value, "unit": "QPs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "C2 fwd+bwd: batch=%d nz=%d nineq=%d neq=%d per GPU, dense random QP "
                                   "(prof-linear.py generator), QPFunction(verbose=-1) defaults" % (B, n, m, q),
                       "global_batch": world * B, "parallelism": "batch-sharded x%d" % world,
                       "ipm_iterations_mean": iters_mean},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "kernel_ms": {"pre_factor": t_pre * 1e3, "ipm": t_ipm * 1e3, "backward": t_bwd * 1e3,
                          "ipm_all_20_iterations": None if t_ipm_fixed is None else t_ipm_fixed * 1e3},
            "job_hbm_roofline_frac": value / world * (fwd_r + fwd_w + bwd_r + bwd_w) / HBM_PEAK,
        }
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
