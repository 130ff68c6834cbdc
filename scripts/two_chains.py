#!/usr/bin/env python3
"""Can the GPU run two latency chains of the large-QP family side by side?  Same box, same process:
  single   one batch of B QPs on one stream (knob: one part, mat-vec in the caller's stream)
  threads  TWO batches of B QPs, each enqueued by its own host thread on its own stream, started together
  library  one batch of 2B QPs, the library's own two parts on two streams (one host thread enqueues both)
Reported: wall time per step (pre-factorisation + loop + backward) with the host synchronised before and after `reps` steps.

    python scripts/two_chains.py [B n m]       default 64 500 500
"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd import _lib  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402

ONE_PART = (1 << 16) | (1 << 26)


def make(B, n, m, seed, dev):
    arrs = problems.prof_qp(B, n, m, 0, seed, np.float64)
    t = [torch.tensor(a, device=dev) if a is not None and a.size else torch.empty(0, dtype=torch.float64, device=dev) for a in arrs]
    return t, torch.ones(B, n, dtype=torch.float64, device=dev)


def step(data, ones):
    Q, p, G, h, A, b = data
    fac = KKTFactors.build(Q, G, A)
    res = fac.ipm(p, h, b)
    return fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=(False, True, False, False, False, False))[1]


def main():
    B, n, m = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (64, 500, 500)
    dev = torch.device("cuda", 0)
    dll = _lib.hip().dll
    reps = 10
    d1, o1 = make(B, n, m, 0, dev)
    d2, o2 = make(B, n, m, 1, dev)
    dboth = [torch.cat([x, y]) if x.nelement() else x for x, y in zip(d1, d2)]
    oboth = torch.cat([o1, o2])

    def run_single(knob, data, ones, stream, out, key):
        dll.qpx_set_ipm_variant(knob)             # the knob is per host thread
        with torch.cuda.stream(stream):
            for _ in range(3):
                step(data, ones)
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                step(data, ones)
            stream.synchronize()
            out[key] = (time.perf_counter() - t0) / reps * 1e3

    for rnd in range(3):
        out = {}
        s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        run_single(ONE_PART, d1, o1, s1, out, "single")
        start = threading.Barrier(2)

        def worker(data, ones, stream, key):
            dll.qpx_set_ipm_variant(ONE_PART)
            with torch.cuda.stream(stream):
                for _ in range(3):
                    step(data, ones)
                stream.synchronize()
                start.wait()
                t0 = time.perf_counter()
                for _ in range(reps):
                    step(data, ones)
                stream.synchronize()
                out[key] = (time.perf_counter() - t0) / reps * 1e3

        th = [threading.Thread(target=worker, args=(d1, o1, s1, "thread a")), threading.Thread(target=worker, args=(d2, o2, s2, "thread b"))]
        [t.start() for t in th]
        [t.join() for t in th]
        run_single(0, dboth, oboth, s1, out, "library")
        run_single(ONE_PART, dboth, oboth, s1, out, "one part")
        print("B=%d n=%d m=%d   single %.3f ms | two threads, two streams: %.3f / %.3f ms | 2B by the library's two parts %.3f ms | 2B as one part %.3f ms"
              % (B, n, m, out["single"], out["thread a"], out["thread b"], out["library"], out["one part"]), flush=True)


if __name__ == "__main__":
    main()
