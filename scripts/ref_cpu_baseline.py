#!/usr/bin/env python3
"""The reference's own CPU PDIPM timed on the headline workload (missing #5).

Runs in the BUILD container only (the GPU box has no /root/reference): imports the unmodified locuslab/qpth from
/root/reference (with an empty in-memory `cvxpy` module, as tests/golden/make_golden.py does -- nothing of cvxpy is on the
PDIPM path) and times what /root/reference/prof-linear.py:95-123 times:

    z = QPFunction(verbose=-1)(Q, p, G, h, A, b); z.backward(ones)          p requires grad (prof-linear.py:99)

at BASELINE.json configs[1] extended with the backward (C2: batch 512, nz = nineq = 100, neq = 0), same generator and seed
as bench.py (tests/problems.py: prof_qp, seed 0), float64 and float32, torch.set_num_threads(all CPUs of the host), one
warm-up, best and median of `--reps` repetitions by time.perf_counter.  Writes one JSON object:

    python scripts/ref_cpu_baseline.py > profiles/ref_cpu_baseline.json

bench.py quotes `qps` of this file (with the host it was measured on) beside its own cpu_baseline leg.
"""
import argparse
import json
import os
import platform
import sys
import time
import types

import numpy as np

sys.modules.setdefault("cvxpy", types.ModuleType("cvxpy"))
sys.path.insert(0, "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
from qpth.qp import QPFunction  # noqa: E402   (the reference's)
import problems  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--nz", type=int, default=100)
    ap.add_argument("--nineq", type=int, default=100)
    ap.add_argument("--neq", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    B, n, m, q = args.batch, args.nz, args.nineq, args.neq
    out = {"what": "the reference's QPFunction forward + backward on the PyTorch CPU backend (unmodified /root/reference, "
                   "prof-linear.py:95-123), generator tests/problems.py:prof_qp seed 0",
           "config": [B, n, m, q], "threads": args.threads, "host_cpus": os.cpu_count(), "cpu": cpu_model(),
           "torch": torch.__version__, "reps": args.reps, "results": {}}
    for name, np_dt in (("f64", np.float64), ("f32", np.float32)):
        Q, p, G, h, A, b = [torch.tensor(x) for x in problems.prof_qp(B, n, m, q, 0, np_dt)]
        if q == 0:
            A = b = torch.empty(0, dtype=Q.dtype)
        p.requires_grad_(True)
        ones = torch.ones(B, n, dtype=Q.dtype)
        ts, tf = [], []
        for rep in range(args.reps + 1):
            t0 = time.perf_counter()
            z = QPFunction(verbose=-1)(Q, p, G, h, A, b)
            t1 = time.perf_counter()
            z.backward(ones)
            t2 = time.perf_counter()
            p.grad = None
            if rep:
                ts.append(t2 - t0)
                tf.append(t1 - t0)
        out["results"][name] = {"qps": B / min(ts), "qps_median": B / float(np.median(ts)), "fwd_bwd_s_best": min(ts),
                                "forward_s_best": min(tf), "forward_only_qps": B / min(tf)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
