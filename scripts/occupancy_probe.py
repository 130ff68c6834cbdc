#!/usr/bin/env python3
"""How the tile LOOP kernel's time answers to the workgroups a CU holds: the same launch with unused LDS added through the
library's measurement hook (QPX_LDS_PAD_BYTES, read once per process -- so one process per value).
    occupancy_probe.py [B n m q]                  parent: runs itself once per pad value
    occupancy_probe.py --one B n m q              child: prints the kernel times under the environment's pad"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(B, n, m, q):
    import torch
    sys.path.insert(0, ROOT)
    from bench import device_batch, time_launches
    from qpth_amd.kkt import KKTFactors
    dev = torch.device("cuda:0")
    if os.environ.get("QPX_VARIANT"):
        from qpth_amd import _lib
        _lib.hip().dll.qpx_set_ipm_variant(int(os.environ["QPX_VARIANT"]))
    Q, p, G, h, A, b = device_batch(B, n, m, dev, torch.float64, 7)
    fac = KKTFactors.build(Q, G, A, B)
    res = fac.ipm(p, h, b)
    torch.cuda.synchronize()
    t_ipm = time_launches(lambda: fac.ipm(p, h, b), 5)
    ones = torch.ones(B, n, dtype=torch.float64, device=dev)
    want_p = (False, True, False, False, False, False)
    t_bwd = time_launches(lambda: fac.backward(res.zhat, res.lam, res.slacks, res.nu, ones, want=want_p), 5)
    t_pre = time_launches(lambda: KKTFactors.build(Q, G, A, B), 5)
    print("pad %6s B: loop %.3f ms   (pre-factorisation %.3f, backward %.3f -- not padded)   iterations mean %.2f" % (
        os.environ.get("QPX_LDS_PAD_BYTES", "0"), t_ipm * 1e3, t_pre * 1e3, t_bwd * 1e3, res.iters.float().mean().item()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(*[int(x) for x in sys.argv[2:6]])
    else:
        dims = sys.argv[1:5] if len(sys.argv) > 4 else ["65536", "64", "64", "0"]
        pads = os.environ.get("PADS", "0 4000 17000 44000 90000").split()
        print("B n m q = %s; QPX_VARIANT=%s" % (" ".join(dims), os.environ.get("QPX_VARIANT", "0")), flush=True)
        for pad in pads:
            env = dict(os.environ, QPX_LDS_PAD_BYTES=pad)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"] + dims, env=env, capture_output=True, text=True)
            sys.stdout.write("".join(l for l in out.stdout.splitlines(True) if l.startswith("pad")) or out.stderr[-800:])
            sys.stdout.flush()
