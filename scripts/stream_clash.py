#!/usr/bin/env python3
"""Does the large-QP family's side stream land on the caller's hardware queue?  HIP deals streams to the device's four hardware
queues in creation order; a side stream on the caller's queue serialises the two parts of a batch (2 x the chain of one part).
One process per case: `k` other streams are created and used once first (shifting the deal), then the caller's stream, then the library
creates its side stream on first use -- and checks that it runs beside the caller's (qpx_hip_api.hip: runs_beside; before that
check k = 6 gave 24.9 ms instead of 11.5).  Prints the step time of C4 (128 QPs, nz = nineq = 500) on that caller stream.

    python scripts/stream_clash.py k [default-stream]
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


def main():
    k = int(sys.argv[1])
    dev = torch.device("cuda", 0)
    B, n, m = 128, 500, 500
    arrs = problems.prof_qp(B, n, m, 0, 0, np.float64)
    data = [torch.tensor(a, device=dev) if a is not None and a.size else torch.empty(0, dtype=torch.float64, device=dev) for a in arrs]
    ones = torch.ones(B, n, dtype=torch.float64, device=dev)
    idle = [torch.cuda.Stream(dev) for _ in range(k)]
    for st in idle:                       # a stream takes its hardware queue with its first piece of work
        with torch.cuda.stream(st):
            torch.zeros(1024, device=dev).add_(1.0)
    torch.cuda.synchronize()
    s = torch.cuda.current_stream(dev) if len(sys.argv) > 2 else torch.cuda.Stream(dev)

    def step():
        Q, p, G, h, A, b = data
        fac = KKTFactors.build(Q, G, A)
        res = fac.ipm(p, h, b)
        return fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=(False, True, False, False, False, False))[1]

    with torch.cuda.stream(s):
        for _ in range(3):
            step()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        s.synchronize()
        print("other streams used first: %d   caller = %s   step %.3f ms"
              % (k, "torch's default stream" if len(sys.argv) > 2 else "a new stream", (time.perf_counter() - t0) / 10 * 1e3), flush=True)


if __name__ == "__main__":
    main()
