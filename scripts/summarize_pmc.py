#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs (FETCH_SIZE / WRITE_SIZE) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, "prof_pmc_" + cname, "**", "*counter_collection*.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != cname:
                    continue
                k = row.get("Kernel_Name", "?")[:60]
                acc[k][0] += float(row.get("Counter_Value", 0))
                acc[k][1] += 1
    print("--", cname, "(sum over dispatches, count, mean per dispatch; units as rocprofv3 reports: KB)")
    for k, (v, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
        print("   %-60s %14.1f %6d %14.2f" % (k, v, c, v / max(c, 1)))
