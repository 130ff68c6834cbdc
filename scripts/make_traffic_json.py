#!/usr/bin/env python3
"""profiles/ipm_traffic.json from the two PMC summaries of scripts/gpu_check.sh:
    make_traffic_json.py <tag>_pmc_FETCH_SIZE.txt <tag>_pmc_WRITE_SIZE.txt [B n m q dtype] > ipm_traffic.json
HBM bytes per launch of the PDIPM loop kernel = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; the factor 2 is the
gfx950 correction of MI355X_MICROARCH.md, "HBM" -- calibrated in round 3 for the 8-byte-per-lane buffer loads the tile
kernels use: 2 GiB streamed once reads FETCH_SIZE = 1.0486e6 KB with raw_buffer_load_b64 and with 16-byte global loads
alike, profiles/archive/r03a_probes_and_phases.txt).  The record carries the digest of the kernel sources it was
measured on; bench.py reports `roofline.traffic` only when that digest is the running build's."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


LOOP = ("k_ipm_tile", "k_ipm_grid", "k_ipm<")
OTHERS = {"pre_factor": ("k_prefac_tile", "k_sweep"), "backward": ("k_kkt_tile", "k_kkt_grid")}


def loop_kernel_mean(path, counter, names=LOOP, must=True):
    best = None
    for line in open(path):
        if any(nm in line for nm in names) and counter in line:
            f = line.split()
            i = f.index(counter)
            n, mean = int(f[i + 1]), float(f[i + 2])
            if best is None or n > best[0]:
                best = (n, mean, " ".join(f[:i]))
    if best is None and must:
        raise SystemExit("no loop-kernel row with %s in %s" % (counter, path))
    return best


def main():
    fetch, write = sys.argv[1], sys.argv[2]
    cfg = sys.argv[3:8] if len(sys.argv) >= 8 else ["512", "100", "100", "0", "f64"]
    nf, f_kb, kname = loop_kernel_mean(fetch, "FETCH_SIZE")
    nw, w_kb, _ = loop_kernel_mean(write, "WRITE_SIZE")
    from bench import kernel_source_digest
    others = {}
    for key, names in OTHERS.items():
        a, b = loop_kernel_mean(fetch, "FETCH_SIZE", names, False), loop_kernel_mean(write, "WRITE_SIZE", names, False)
        if a and b:
            others[key] = {"hbm_bytes_per_launch": int(round((2.0 * a[1] + b[1]) * 1024)), "fetch_size_kb": a[1], "write_size_kb": b[1],
                           "launches": [a[0], b[0]], "kernel": a[2]}
    rec = {"config": [int(cfg[0]), int(cfg[1]), int(cfg[2]), int(cfg[3]), cfg[4]],
           "hbm_bytes_per_launch": int(round((2.0 * f_kb + w_kb) * 1024)),
           "fetch_size_kb": f_kb, "write_size_kb": w_kb, "launches": [nf, nw], "kernel": kname,
           "other_kernels": others,
           "kernel_source_digest": kernel_source_digest(),
           "fetch_size_factor": 2.0,
           "fetch_size_factor_source": "profiles/archive/r03a_probes_and_phases.txt: 2 GiB streamed once by raw_buffer_load_b64 (8 B per lane) -> FETCH_SIZE 1048603 KB = 0.500 of the bytes; same for 16 B per lane",
           "source": "%s, %s: 2*FETCH_SIZE + WRITE_SIZE, KB -> bytes" % (os.path.basename(fetch), os.path.basename(write))}
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
