#!/usr/bin/env python3
"""Eager launches against the replay of one captured hipGraph, same box, same process: pre-factorisation + PDIPM loop +
backward (p-gradient) through KKTFactors (the stream-ordered C ABI underneath; QPFunction itself reads the pre-factorisation's
status word back to raise the reference's errors and is therefore not capturable).

    python scripts/graph_replay.py [B n m q ...]      default: C2, C3, C4 and three small large-QP batches
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from qpth_amd.kkt import KKTFactors  # noqa: E402


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    nums = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(nums[i:i + 4]) for i in range(0, len(nums), 4)] or [
        (512, 100, 100, 0), (512, 100, 50, 10), (128, 500, 500, 0), (16, 300, 200, 20), (8, 500, 500, 0), (1, 500, 500, 0)]
    for B, n, m, q in shapes:
        arrs = problems.prof_qp(B, n, m, q, 0, np.float64)
        Q, p, G, h, A, b = [torch.tensor(a, device=dev) if a is not None and a.size else torch.empty(0, dtype=torch.float64, device=dev)
                            for a in arrs]
        ones = torch.ones(B, n, dtype=torch.float64, device=dev)
        want = (False, True, False, False, False, False)

        def step():
            fac = KKTFactors.build(Q, G, A)
            res = fac.ipm(p, h, b)
            return res.zhat, fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=want)[1]

        for _ in range(3):
            z0, g0 = step()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            z1, g1 = step()
        graph.replay()
        torch.cuda.synchronize()
        same = torch.equal(z0, z1) and torch.equal(g0, g1)
        reps = 50 if B * n < 40000 else 20
        rows = []
        for _ in range(3):
            rows.append((timed(step, reps), timed(graph.replay, reps)))
        e, g = min(r[0] for r in rows), min(r[1] for r in rows)
        print("B n m q = %4d %3d %3d %3d   eager %8.4f ms   graph replay %8.4f ms   (%+.1f %%)   replay == eager: %s"
              % (B, n, m, q, e, g, (g / e - 1) * 100, same), flush=True)


if __name__ == "__main__":
    main()
