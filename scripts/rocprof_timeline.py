#!/usr/bin/env python3
"""Per-dispatch view of a rocprofv3 --kernel-trace database (rocpd sqlite, ROCm 7.2): the dispatches of the LAST
`--last N` qpx kernels grouped by (kernel, grid), with count / mean / total duration, the idle time between
consecutive dispatches and the wall time the window spans (kernels on different streams overlap, so the sum of the
durations can exceed it).  Usage: rocprof_timeline.py <results.db> [--last N] [--list]"""
import sqlite3
import sys


def short(name):
    name = name.replace("void qpx::", "").replace("qpx::", "")
    return name.split("(")[0][:44]


def main():
    path = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 500
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, start, end, grid_x, grid_y, workgroup_x, queue_id from kernels order by start"))
    rows = [r for r in rows if "qpx::" in r[0]][-last:]
    if not rows:
        print("no qpx kernels in", path)
        return
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    print("## %s: last %d qpx dispatches, window %.1f us, sum of durations %.1f us, queues used: %s" % (
        path, len(rows), (t1 - t0) / 1e3, sum(r[2] - r[1] for r in rows) / 1e3, sorted(set(r[6] for r in rows))))
    groups = {}
    order = []
    for name, s, e, gx, gy, wx, q in rows:
        k = (short(name), gx // max(wx, 1), gy)
        if k not in groups:
            groups[k] = []
            order.append(k)
        groups[k].append((e - s) / 1e3)
    print("%-46s %12s %6s %10s %10s %10s" % ("kernel", "grid", "n", "mean_us", "max_us", "total_us"))
    for k in sorted(order, key=lambda k: -sum(groups[k])):
        d = groups[k]
        print("%-46s %12s %6d %10.1f %10.1f %10.1f" % (k[0], "%dx%d" % (k[1], k[2]), len(d), sum(d) / len(d), max(d), sum(d)))
    # busy time = union of the dispatch intervals
    busy, cur_s, cur_e = 0, None, None
    for name, s, e, *_ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("GPU busy (union of dispatches) %.1f us = %.1f %% of the window" % (busy / 1e3, 100.0 * busy / (t1 - t0)))
    if "--list" in sys.argv:
        prev = None
        for name, s, e, gx, gy, wx, q in rows:
            print("%-40s q%-2d %5dx%-3d start %10.1f dur %8.1f gap %7.1f" % (
                short(name), q, gx // max(wx, 1), gy, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
            prev = e


if __name__ == "__main__":
    main()
