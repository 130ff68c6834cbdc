#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X, and the reference's two timing tables.

    python bench.py --gpus N --steps K --warmup W                    (the driver's contract)
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic QPs that is already resident in HBM:
    z = QPFunction(verbose=-1)(Q, p, G, h, A, b); z.backward(ones)        (forward + backward)
the measurement of SURVEY.md section 8d / prof-linear.py:110-118, device-synchronised, `p` requiring grad as in
prof-linear.py:99.  Default workload = BASELINE.json configs[1] extended with the backward pass as its `metric`
asks: batch=512, nz=100, nineq=100, neq=0, float64 (the dtype the 1e-4 parity gate holds in).

Before the W warm-up steps the script runs --spin-up seconds (default 0.5) of the same step, untimed: the first GPU work of a
fresh box runs ~7 % slower for its first few hundred steps (774 K against 831 K QPs/s, profiles/r05v_spin_up.txt); the timed
region is exactly K steps either way (`untimed_spin_up_s` in the JSON line; --spin-up 0 turns it off).  So that rounds stay
comparable the line also carries the number of the protocol of rounds 1-4 from the same process: W warm-up steps, then 20
timed steps, BEFORE the spin-up (`first_20_steps_qps`), and where the wall time of the run went (`wall_s`).

QPX_FORCE_DIST=1 takes the N > 1 branch (RCCL process group, zhat all_gathered beside the backward, the C5 point) with
WORLD_SIZE = 1: the collective code path on the one GPU of a box (tests/test_gpu_dist.py).

Multi-GPU (one process per GPU, RCCL):
  --config c2 (default)  every rank solves its own 512-QP shard and the ranks all_gather zhat (the only exchange
                         the path has, north_star / SURVEY 8e) inside the timed region  => "scaling": "weak"
  --config c5            BASELINE.json configs[4]: a FIXED global batch of 65 536 QPs (nz=nineq=64) split by
                         qpth_amd.dist.shard_bounds, zhat gathered                       => "scaling": "strong"

The JSON line also carries
  roofline ...... the dominant kernel (the PDIPM loop): algorithmic flops and bytes per launch / the launch's
                  duration measured live with HIP events on the launch stream; `traffic` = HBM bytes per launch
                  from the PMC passes of scripts/gpu_check.sh, reported only if they were taken on THIS build
  cpu_baseline .. the oracle (C restatement of the reference's CPU path, OpenMP over QPs) timed on this host
  fwd_only ...... QPs/s of the forward alone (BASELINE.json configs[1] as written)
  kernel_ms ..... per-launch HIP-event times, incl. the loop with early stopping off (all maxIter iterations =
                  the work the reference does) and the backward with every gradient requested

Tables of the reference's timing scripts (one JSON line per row, after the headline line is suppressed):
  --table prof-linear   prof-linear.py:38-61: batch 128, nz = nineq in {10, 50, 100, 500}, forward and backward
                        timed separately, f32 and f64
  --table prof-gurobi   prof-gurobi.py:37-48,115-118: nz=100 (dense Q), nineq=100, batch in {1, 64, 128},
                        pre_factor_kkt + forward only
"""
import argparse
import ctypes
import hashlib
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
from qpth_amd import _lib, dist as qdist  # noqa: E402

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
# Dense f64 / f32 peaks.  f32: 157.3 TF (MI355X_MICROARCH.md, matrix = vector).  f64: the guide has no
# row for it; 78.6 TF is AMD's MI355X datasheet number for both the vector and the matrix pipe, and our
# micro-benchmarks agree: v_fma_f64 issues every 4.07 ticks per wave (77 TF over 1024 SIMDs at the
# measured 2.39 GHz tick), v_mfma_f64_16x16x4 every 83 ticks (60 TF) -- scripts/ubench_mfma.py.
MFMA_PEAK = {"f64": 78.6e12, "f32": 157.3e12}
# The reference's own PyTorch-CPU PDIPM on this workload cannot be re-timed on the GPU box (no /root/reference there): it
# is timed in the build container by scripts/ref_cpu_baseline.py (the unmodified reference, all host threads), whose
# record -- profiles/ref_cpu_baseline.json: QPs/s, thread count, CPU model -- is quoted next to the port that can run here.
def reference_cpu_record(dtype):
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "ref_cpu_baseline.json")))
        r = rec["results"][dtype]
        return {"qps": r["qps"], "threads": rec["threads"], "cpu": rec["cpu"], "config": rec["config"],
                "source": "profiles/ref_cpu_baseline.json (scripts/ref_cpu_baseline.py, build container)"}
    except Exception as e:        # a missing or broken record is reported, never replaced by a constant
        return {"qps": None, "source": "profiles/ref_cpu_baseline.json unreadable: %s" % e}


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


def kernel_source_digest():
    """sha256 over the kernel sources with comments and white space removed: identifies the CODE a PMC measurement
    belongs to (a git commit would change with every documentation edit, the raw files with every comment)."""
    import re
    hsh = hashlib.sha256()
    d = os.path.join(ROOT, "qpth_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name == "qpx_bench.hip":       # the micro-benchmarks (libqpx_bench.so): not part of the product library
            continue
        if name.endswith((".h", ".hip", ".inc")) or name == "Makefile":
            text = open(os.path.join(d, name), "r", errors="replace").read()
            if name != "Makefile":
                text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
                text = re.sub(r"//[^\n]*", " ", text)
            else:
                text = re.sub(r"#[^\n]*", " ", text)
            hsh.update(name.encode())
            hsh.update(" ".join(text.split()).encode())
    return hsh.hexdigest()[:16]


def time_launches(fn, nrep):
    """average duration of `fn`'s launches by HIP events on the launch stream (torch's current stream)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(nrep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nrep * 1e-3


def make_batch(B, n, m, q, seed, np_dt, dev):
    Q, p, G, h, A, b = problems.prof_qp(B, n, m, q, seed=seed, dtype=np_dt)
    return (Q, p, G, h, A, b), [torch.tensor(x, device=dev) for x in (Q, p, G, h, A, b)]


# ---------------------------------------------------------------------------------------------- tables
def table_prof_linear(dev, args):
    """prof-linear.py:38-61,95-123: nBatch = 128, nz = nineq in {10, 50, 100, 500}, neq = 0; forward and
    backward timed separately (device-synchronised), p requires grad; both dtypes."""
    rows = []
    for dtype, np_dt in (("f32", np.float32), ("f64", np.float64)):
        for nz in (10, 50, 100, 500):
            B = 128
            _, (tQ, tp, tG, th, tA, tb) = make_batch(B, nz, nz, 0, 0, np_dt, dev)
            tp.requires_grad_(True)
            ones = torch.ones(B, nz, dtype=tQ.dtype, device=dev)
            qpf = QPFunction(verbose=-1)
            ntr = 3 if nz >= 500 else 10
            fw, bw = [], []
            for i in range(ntr + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                z = qpf(tQ, tp, tG, th, tA, tb)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                z.backward(ones)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                tp.grad = None
                if i:
                    fw.append(t1 - t0)
                    bw.append(t2 - t1)
            # what the kernels computed in: float32 tensors at the sizes the float64 tile kernels serve run in float64
            # arithmetic (QPFunction(refine=None)); elsewhere the float32 kernels + two finishing iterations
            from qpth_amd.qp import f64_arithmetic_serves
            arith = "f64" if dtype == "f64" else ("f64 (QPX_F32_WIDE: float32 tensors, float64 factors and arithmetic)"
                                                  if f64_arithmetic_serves(nz, nz, 0, _lib.hip()) else "f32 + 2 finishing iterations in f64 residuals")
            rows.append({"table": "prof-linear", "dtype": dtype, "arithmetic": arith, "nBatch": B, "nz": nz, "nineq": nz, "neq": 0,
                         "forward_ms": float(np.median(fw)) * 1e3, "backward_ms": float(np.median(bw)) * 1e3,
                         "qps_fwd_bwd": B / (float(np.median(fw)) + float(np.median(bw))), "trials": ntr})
            print(json.dumps(rows[-1]), flush=True)
    return rows


def table_prof_gurobi(dev, args):
    """prof-gurobi.py:37-48,109-118: nz = 100, nineq = 100, neq = 0, nBatch in {1, 64, 128}; the timed region is
    pre_factor_kkt + forward only (no backward), f64 as in the script (`.double()`)."""
    from qpth_amd.solvers.pdipm import batch as pdipm_b
    rows = []
    for B in (1, 64, 128):
        _, (tQ, tp, tG, th, tA, tb) = make_batch(B, 100, 100, 0, 0, np.float64, dev)
        ts = []
        for i in range(11):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(tQ, tG, tA)
            pdipm_b.forward(tQ, tp, tG, th, tA, tb, Q_LU, S_LU, R, verbose=-1)
            torch.cuda.synchronize()
            if i:
                ts.append(time.perf_counter() - t0)
        rows.append({"table": "prof-gurobi", "dtype": "f64", "arithmetic": "f64", "nBatch": B, "nz": 100, "nineq": 100, "neq": 0,
                     "pre_factor_plus_forward_ms": float(np.median(ts)) * 1e3, "qps_forward": B / float(np.median(ts))})
        print(json.dumps(rows[-1]), flush=True)
    return rows


def c5_point(rank, world, dev, dtype, barrier, steps=5, warmup=2):
    """The north star's strong-scaling point on `world` GPUs: BASELINE.json configs[4] (global batch 65 536, nz = nineq = 64,
    neq = 0), this rank's contiguous slice generated on the device from a per-rank seed (prof-linear.py's generator in
    torch), fwd+bwd with zhat all_gathered beside the backward, barrier + synchronize on both sides, MAX over ranks."""
    import torch.distributed as dist
    GB, n, m = 65536, 64, 64
    lo, hi = qdist.shard_bounds(GB, rank, world)
    B = hi - lo
    Q, p, G, h, e, _ = device_batch(B, n, m, dev, dtype, 1000 + rank)
    p.requires_grad_(True)
    ones = torch.ones(B, n, dtype=dtype, device=dev)
    qpf = QPFunction(verbose=-1)

    def step():
        z = qpf(Q, p, G, h, e, e)
        pending = qdist.gather_batch(z.detach(), GB, async_op=True)
        z.backward(ones)
        pending.wait()
        p.grad = None

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    return {"metric": "QPs/sec (fwd+bwd), C5: GLOBAL batch 65536 nz=64 nineq=64 neq=0 sharded over the GPUs, zhat all_gathered",
            "value": GB * steps / dt, "unit": "QPs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "scaling": "strong", "global_batch": GB, "per_gpu_batch": B,
            "data": "synthetic, generated on the device"}


def device_batch(B, n, m, dev, dtype, seed):
    """prof-linear.py:64-75's generator in torch on the device (neq = 0): 65 536 QPs of C5 are 4.3 GB of Q and G, which the
    numpy generator + H2D copy would spend half a minute on"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    L = torch.rand(B, n, n, dtype=dtype, device=dev, generator=gen)
    Q = L @ L.transpose(1, 2) + 1e-3 * torch.eye(n, dtype=dtype, device=dev)
    del L
    G = torch.randn(B, m, n, dtype=dtype, device=dev, generator=gen)
    z0 = torch.randn(B, n, dtype=dtype, device=dev, generator=gen)
    h = torch.bmm(G, z0.unsqueeze(2)).squeeze(2) + torch.rand(B, m, dtype=dtype, device=dev, generator=gen)
    p = torch.randn(B, n, dtype=dtype, device=dev, generator=gen)
    e = torch.empty(0, dtype=dtype, device=dev)
    return Q, p, G, h, e, e


def side_config(dev, B, n, m, q, steps, warmup, on_device=False):
    """One of BASELINE.json's other configurations timed the same way as the headline (K steps of fwd+bwd bracketed by
    synchronize, float64, p requires grad), after the headline's timed region -- so that the record of the default
    command (the one the driver runs) also holds C3 and C4, which were builder-run numbers only until round 5."""
    if on_device:
        tQ, tp, tG, th, tA, tb = device_batch(B, n, m, dev, torch.float64, 1000)
    else:
        _, (tQ, tp, tG, th, tA, tb) = make_batch(B, n, m, q, 0, np.float64, dev)
    tp.requires_grad_(True)
    ones = torch.ones(B, n, dtype=tQ.dtype, device=dev)
    qpf = QPFunction(verbose=-1)

    def step():
        z = qpf(tQ, tp, tG, th, tA, tb)
        z.backward(ones)
        tp.grad = None

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "fwd+bwd: batch=%d nz=%d nineq=%d neq=%d, f64" % (B, n, m, q), "value": B * steps / dt, "unit": "QPs/s",
            "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup}


# ---------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spin-up", type=float, default=0.5, help="seconds of untimed steps in front of the warm-up steps (0 = none)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5", "custom"],
                    help="c2: B=512 nz=100 nineq=100 per GPU (default, weak scaling); c3: B=512 nz=100 nineq=50 "
                         "neq=10 per GPU; c4: B=128 nz=nineq=500 per GPU; c5: fixed GLOBAL batch 65536, nz=nineq=64 "
                         "(strong scaling); custom: --batch/--nz/--nineq/--neq per GPU")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--nz", type=int, default=None)
    ap.add_argument("--nineq", type=int, default=None)
    ap.add_argument("--neq", type=int, default=None)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave zhat on its rank (no all_gather)")
    ap.add_argument("--shared", action="store_true",
                    help="Q, G, A shared by the batch (SURVEY 8f-1): one pre-factorisation for the whole batch, "
                         "shared-parameter gradients reduced by qpx_batch_outer (+ all_reduce at N > 1)")
    ap.add_argument("--refine", type=int, default=None,
                    help="QPFunction(refine=...).  float32 tensors: default = float64 arithmetic where the float64 tile "
                         "kernels serve the size (else 2); 0 = the float32 loop kernel alone; k = float32 kernels + k "
                         "finishing iterations on the residuals of the original data.  float64: default 0")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="default command only: do not time BASELINE.json's C3 and C4 behind the headline (extra.other_baseline_configs)")
    ap.add_argument("--step-kernels-only", action="store_true",
                    help="profiling runs: skip the launches a step does not contain (the loop with early stopping off, the backward "
                         "with every gradient, the forward under no_grad), so that per-kernel averages and PMC sums are the step's")
    ap.add_argument("--table", default=None, choices=["prof-linear", "prof-gurobi"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path to time.")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    t_start = time.perf_counter()
    wall = {}
    distributed = world > 1 or os.environ.get("QPX_FORCE_DIST", "") == "1"
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    if os.environ.get("QPX_VARIANT"):            # A/B knob for kernel development (include/qpx.h)
        _lib.hip().dll.qpx_set_ipm_variant(int(os.environ["QPX_VARIANT"]))

    if args.table:
        assert world == 1, "--table runs on one GPU"
        (table_prof_linear if args.table == "prof-linear" else table_prof_gurobi)(dev, args)
        return

    presets = {"c2": (512, 100, 100, 0), "c3": (512, 100, 50, 10), "c4": (128, 500, 500, 0), "c5": (65536, 64, 64, 0)}
    if args.config == "custom" or any(v is not None for v in (args.batch, args.nz, args.nineq, args.neq)):
        base = presets.get(args.config, presets["c2"])
        Bcfg, n, m, q = [v if v is not None else d for v, d in zip((args.batch, args.nz, args.nineq, args.neq), base)]
    else:
        Bcfg, n, m, q = presets[args.config]
    strong = args.config == "c5"
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    w = 8 if args.dtype == "f64" else 4
    if strong:
        # fixed global batch: this rank's contiguous slice, generated per shard from a per-rank seed
        lo, hi = qdist.shard_bounds(Bcfg, rank, world)
        B, global_B = hi - lo, Bcfg
    else:
        B, global_B = Bcfg, Bcfg * world
    host, (tQ, tp, tG, th, tA, tb) = make_batch(B, n, m, q, rank, np_dt, dev)
    if args.shared:
        torch.manual_seed(1234 + rank)        # (the launch lasts as long as its slowest QP: unseeded data moved the line by 10 % between runs)
        tQ, tG = tQ[0].contiguous(), tG[0].contiguous()
        tz0 = torch.randn(B, n, dtype=tQ.dtype, device=dev)
        th = tz0 @ tG.t() + torch.rand(B, m, dtype=tQ.dtype, device=dev)
        if q:
            tA = tA[0].contiguous()
            tb = tz0 @ tA.t()
        tQ.requires_grad_(True)
        tG.requires_grad_(True)
    tp.requires_grad_(True)                      # prof-linear.py:99
    ones = torch.ones(B, n, dtype=tQ.dtype, device=dev)
    qpf = QPFunction(verbose=-1, refine=args.refine)
    gather = distributed and not args.no_gather

    def step():
        z = qpf(tQ, tp, tG, th, tA, tb)
        # zhat for the caller's autograd graph: the all_gather runs on RCCL's stream BESIDE the backward launches
        # (the backward needs nothing from the other ranks) and is waited for at the end of the step
        pending = qdist.gather_batch(z.detach(), global_B, async_op=True) if gather else None
        z.backward(ones)
        if pending is not None:
            pending.wait()
        if args.shared and distributed:
            qdist.reduce_shared_grad(tQ.grad, B, global_B)
            qdist.reduce_shared_grad(tG.grad, B, global_B)
        tp.grad = None
        if args.shared:
            tQ.grad = None
            tG.grad = None
        return z

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # the protocol of rounds 1-4, kept so that rounds stay comparable: W warm-up steps, then 20 timed steps, as the process'
    # first GPU work (un-spun)
    for _ in range(args.warmup):
        step()
    barrier()
    u0 = time.perf_counter()
    for _ in range(20):
        step()
    barrier()
    first20 = time.perf_counter() - u0
    if distributed:
        tt = torch.tensor([first20], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        first20 = float(tt.item())
    wall["setup_and_first_20_steps"] = time.perf_counter() - t_start
    # untimed: the device's clocks and the allocator settle (a fresh box runs its first few hundred steps up to 7 % slower than
    # its later ones: 774 K against 831 K QPs/s in profiles/r05v_spin_up.txt; 1-2 % in a process on a box that has already
    # worked); the warm-up proper follows.  The timed region is exactly `steps` steps.
    w0 = time.perf_counter()
    spin_until = time.perf_counter() + args.spin_up
    while True:
        go = time.perf_counter() < spin_until
        if distributed:              # (a step holds collectives at N > 1: every rank must run the same number of them)
            flag = torch.tensor([1.0 if go else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = bool(flag.item() > 0)
        if not go:
            break
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    wall["spin_up_and_warmup"] = time.perf_counter() - w0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    wall["timed_region"] = dt
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = global_B * args.steps / dt

    # median over chunks of 10 steps: the box-to-box and run-to-run spread the single total hides (every rank
    # runs them: a step contains collectives at N > 1)
    w0 = time.perf_counter()
    chunk_ms = []
    for _ in range(min(10, max(3, args.steps // 10))):
        barrier()
        c0 = time.perf_counter()
        for _ in range(10):
            step()
        barrier()
        chunk_ms.append((time.perf_counter() - c0) / 10 * 1e3)

    # N > 1 on the default configuration: the line ALSO carries the north star's strong-scaling point (BASELINE.json
    # configs[4]: C5, fixed GLOBAL batch 65 536, nz = nineq = 64, contiguous slices, zhat all_gathered), so that whichever
    # command the driver runs at N = 1, 2, 4, 8 the record holds both readings of the metric.  Every rank takes part.
    wall["median_chunks"] = time.perf_counter() - w0
    w0 = time.perf_counter()
    extra = None
    if distributed and args.config == "c2" and not args.shared and all(v is None for v in (args.batch, args.nz, args.nineq, args.neq)):
        extra = {"c5_strong_scaling": c5_point(rank, world, dev, tQ.dtype, barrier)}
    if (not distributed and args.config == "c2" and not args.shared and args.dtype == "f64" and args.refine is None
            and not args.no_side_configs and all(v is None for v in (args.batch, args.nz, args.nineq, args.neq))):
        # every BASELINE.json configuration that fits one GPU behind the headline, so that the driver's record holds them:
        # C3, C4 and -- round 6 -- C5's 65 536 QPs on this one GPU (the point the 1/2/4/8 curve starts from)
        extra = {"other_baseline_configs": {"c3": side_config(dev, 512, 100, 50, 10, 100, 10),
                                            "c4": side_config(dev, 128, 500, 500, 0, 10, 2),
                                            "c5_one_gpu": side_config(dev, 65536, 64, 64, 0, 5, 2, on_device=True)}}
    wall["side_configs"] = time.perf_counter() - w0
    w0 = time.perf_counter()

    # float32 data at a size the float64 tile kernels serve runs in float64 arithmetic (QPFunction(refine=None),
    # QPX_F32_WIDE): the kernels timed and priced below are then the float64 ones, reading and writing float32 tensors
    from qpth_amd.qp import f64_arithmetic_serves
    wide = args.dtype == "f32" and args.refine is None and f64_arithmetic_serves(n, m, q, _lib.hip())
    arith = "f64" if wide else args.dtype
    if rank == 0:
        # ---- per-kernel timing with HIP events on the launch stream (torch's current stream) ----
        dQ, dG, dA = tQ.detach(), tG.detach(), tA.detach() if q else tA
        kp, kh, kb, ones_k = tp.detach(), th, tb, ones
        fac = KKTFactors.build(dQ, dG, dA, B, wide=wide)
        res = fac.ipm(kp, kh, kb)
        torch.cuda.synchronize()
        iters = res.iters.cpu().numpy()
        iters_mean = float(iters.mean())
        nrep = 50 if n + m <= 256 else 5
        t_pre = time_launches(lambda: KKTFactors.build(dQ, dG, dA, B, wide=wide), nrep)
        t_ipm = time_launches(lambda: fac.ipm(kp, kh, kb), nrep)
        want_p = (False, True, False, False, False, False)
        t_bwd = time_launches(lambda: fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones_k, want=want_p), nrep)
        t_bwd_all = t_ipm_fixed = t_fwd = None
        if not args.step_kernels_only:
            t_bwd_all = time_launches(lambda: fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones_k), nrep)
            set_stall_policy(_lib.STALL_OFF)
            t_ipm_fixed = time_launches(lambda: fac.ipm(kp, kh, kb), max(3, nrep // 3))
            set_stall_policy(None)
            # forward only = BASELINE.json configs[1] as written (QPFunction forward, no backward), on the caller's tensors
            with torch.no_grad():
                t_fwd = time_launches(lambda: qpf(tQ.detach(), tp.detach(), tG.detach(), th, tA.detach() if q else tA, tb), nrep)

        fwd_r, fwd_w, bwd_r, bwd_w = algorithmic_bytes_per_qp(n, m, q, w)
        ipm_bytes = (fwd_r + fwd_w) * B          # the forward's compulsory traffic, DESIGN.md section 6
        # The loop kernel is compute-side bound (48 flop per compulsory byte at C2, machine balance ~10):
        # its roofline is the dense matrix/vector peak of the dtype it computes in.
        ipm_flops = algorithmic_flops_per_qp(n, m, q, iters_mean) * B
        achieved = ipm_flops / t_ipm
        peak = MFMA_PEAK[arith]
        traffic, traffic_note = None, "no PMC record for this configuration"
        other_traffic = {}
        tf = os.path.join(ROOT, "profiles", "ipm_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                if rec.get("config") != [B, n, m, q, arith]:
                    traffic_note = "PMC record is for config %s" % rec.get("config")
                elif rec.get("kernel_source_digest") != kernel_source_digest():
                    traffic_note = "PMC record is of another build (%s); re-run scripts/gpu_check.sh" % rec.get("kernel_source_digest")
                else:
                    traffic, traffic_note = rec.get("hbm_bytes_per_launch"), rec.get("source")
                    other_traffic = rec.get("other_kernels", {})
            except Exception as e:                      # a broken record is reported, never silently reused
                traffic_note = "unreadable PMC record: %s" % e
        roofline = {"kernel": "PDIPM loop kernel (k_ipm_tile / k_ipm_grid, one launch per forward)", "bound": "mfma",
                    "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic, "traffic_source": traffic_note,
                    "algorithmic_flops_per_launch": ipm_flops,
                    "algorithmic_bytes_per_launch": ipm_bytes, "hbm_frac": ipm_bytes / t_ipm / HBM_PEAK,
                    "launch_ms": t_ipm * 1e3, "kernel_source_digest": kernel_source_digest()}
        # the two other launches of a step, both HBM-side: bytes they have to move (SURVEY 8d's backward reads + the one
        # gradient the benchmark asks for; the pre-factorisation's reads of Q, G, A + the factor blob it exists to write)
        # against the bytes the counters saw (same PMC passes, same digest rule as `traffic`)
        pre_bytes = (w * (n * n + m * n + q * n) + fac.elems * fac.blob.element_size()) * B
        bwd_bytes = (bwd_r + w * n) * B
        roofline["other_kernels"] = {
            "pre_factor": {"bound": "hbm", "algorithmic_bytes_per_launch": pre_bytes, "launch_ms": t_pre * 1e3,
                           "achieved": pre_bytes / t_pre / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": pre_bytes / t_pre / HBM_PEAK,
                           "traffic": (other_traffic.get("pre_factor") or {}).get("hbm_bytes_per_launch")},
            "backward": {"bound": "hbm", "algorithmic_bytes_per_launch": bwd_bytes, "launch_ms": t_bwd * 1e3,
                         "achieved": bwd_bytes / t_bwd / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bwd_bytes / t_bwd / HBM_PEAK,
                         "traffic": (other_traffic.get("backward") or {}).get("hbm_bytes_per_launch")}}

        # Large-QP family (C4): a forward is ~400 stream-ordered launches; the dominant kernel is the MFMA tile GEMM
        # (k_big_gemm, ~70 % of the time).  Its largest launch, R = Zt Zt^T of the pre-factorisation, is re-issued
        # alone through the measurement hook and priced against the f64/f32 matrix-core peak: m^2 n flops per QP
        # (one triangle of the symmetric product), Zt read + R written per QP.
        lib = _lib.hip()
        code = _lib.QPX_F32_WIDE if wide else (_lib.QPX_F64 if arith == "f64" else _lib.QPX_F32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        blob_ptr = ctypes.c_void_p(fac.blob.data_ptr())
        with fac._knob():
            is_big = lib.dll.qpx_big_gemm_r(code, B, n, m, q, blob_ptr, ctypes.c_void_p(stream)) == 0
        if is_big:
            def gemm_r():
                with fac._knob():
                    lib.check(lib.dll.qpx_big_gemm_r(code, B, n, m, q, blob_ptr, ctypes.c_void_p(stream)))
            t_gemm = time_launches(gemm_r, 20)
            g_flops = float(m) * m * n * B
            g_bytes = float(m * n + m * m) * w * B
            roofline = {"kernel": "k_big_gemm, launch R = Zt Zt^T of the large-QP pre-factorisation (one launch, the whole batch)",
                        "bound": "mfma", "achieved": g_flops / t_gemm / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                        "frac": g_flops / t_gemm / peak, "traffic": None,
                        "traffic_source": "no PMC pass for this launch",
                        "algorithmic_flops_per_launch": g_flops, "algorithmic_bytes_per_launch": g_bytes,
                        "hbm_frac": g_bytes / t_gemm / HBM_PEAK, "launch_ms": t_gemm * 1e3,
                        "kernel_source_digest": kernel_source_digest(),
                        "whole_loop": {"what": "all launches of qpx_ipm together, algorithmic flops of the loop / their time",
                                       "achieved": achieved / 1e12, "frac": achieved / peak}}

        wall["kernel_timing"] = time.perf_counter() - w0
        w0 = time.perf_counter()
        cpu_baseline = None
        if not args.no_cpu_baseline and world == 1 and not args.shared:      # reported at N = 1 only
            from oracle import qp_oracle as orc
            ncpu = os.cpu_count() or 1
            Bs = B if n + m <= 256 else min(B, 32)       # bounded sample: the full C2 batch; 32 QPs at C4 size
            sQ, sp, sG, sh, sA, sb = [x[:Bs] if x.ndim > 1 or x.size else x for x in host]
            np1 = np.ones((Bs, n), np_dt)

            def cpu_pass(nth):
                c0 = time.perf_counter()
                o = orc.OracleQP(sQ, sp, sG, sh, sA, sb, nthreads=nth)
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
            cpu_baseline = {"value": Bs / best, "unit": "QPs/s", "cores": cores, "kind": "port",
                            "sample": "B=%d QPs of the workload, fwd+bwd, best of 3 at the best OpenMP team size of a "
                                      "scan over 4..128 threads, %s, all %d IPM iterations of the reference's batch-global "
                                      "loop (the GPU loop stops each QP when it has converged: %.1f on average), %d host "
                                      "CPUs.  `reference_pytorch_cpu`: the reference's own PyTorch-CPU PDIPM at C2, timed "
                                      "in the build container (it cannot run on the GPU box)"
                                      % (Bs, args.dtype, int(info["trips"]), iters_mean, ncpu),
                            "reference_pytorch_cpu": reference_cpu_record(args.dtype)}

        wall["cpu_baseline"] = time.perf_counter() - w0
        wall["total"] = time.perf_counter() - t_start
        names = {"c2": "C2", "c3": "C3", "c4": "C4", "c5": "C5", "custom": "custom"}
        out = {
            "metric": "QPs/sec (fwd+bwd) at batch=512 nz=100 nineq=100; 1/2/4/8 MI355X",
            "value": value, "unit": "QPs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": arith, "data": "synthetic", "untimed_spin_up_s": args.spin_up,
            "config": {"workload": "%s fwd+bwd: batch=%d nz=%d nineq=%d neq=%d %s, dense random QP (prof-linear.py "
                                   "generator)%s, QPFunction(verbose=-1%s) defaults, p requires grad%s"
                                   % (names[args.config], Bcfg, n, m, q, "GLOBAL (sharded)" if strong else "per GPU",
                                      ", Q G A shared by the batch" if args.shared else "",
                                      "" if args.refine is None else ", refine=%d" % args.refine,
                                      ", zhat all_gathered over RCCL" if gather else ""),
                       "global_batch": global_B, "parallelism": "batch-sharded x%d" % world,
                       "tensor_dtype": args.dtype,
                       "arithmetic": ("float64 kernels on float32 tensors (QPX_F32_WIDE: widened on load, narrowed on store, "
                                      "float64 factors)" if wide else arith),
                       "ipm_iterations_mean": iters_mean, "ipm_iterations_max": int(iters.max()),
                       "ipm_iterations_histogram": np.bincount(iters, minlength=21).tolist()},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "ms_per_step_median_of_10_step_chunks": float(np.median(chunk_ms)),
            "first_20_steps_qps": global_B * 20 / first20,
            "first_20_steps_what": "the protocol of rounds 1-4 in this same process: %d warm-up steps, then 20 timed steps, as the "
                                   "process' first GPU work, before the untimed spin-up" % args.warmup,
            "wall_s": {k: round(v, 3) for k, v in wall.items()},
            "fwd_only": None if t_fwd is None else {"qps": B / t_fwd, "ms": t_fwd * 1e3, "what": "QPFunction forward under no_grad (BASELINE.json configs[1])"},
            "kernel_ms": {"pre_factor": t_pre * 1e3, "ipm": t_ipm * 1e3, "backward": t_bwd * 1e3,
                          "backward_all_gradients": None if t_bwd_all is None else t_bwd_all * 1e3,
                          "ipm_all_20_iterations": None if t_ipm_fixed is None else t_ipm_fixed * 1e3},
            "job_hbm_roofline_frac": value / world * (fwd_r + fwd_w + bwd_r + bwd_w) / HBM_PEAK,
        }
        if extra is not None:
            out["extra"] = extra
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
