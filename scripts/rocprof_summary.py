#!/usr/bin/env python3
"""Text summary of rocprofv3 results (*.db, the rocpd sqlite format ROCm 7.2 writes):
kernel-trace stats (calls / average / total / share) and, for --pmc runs, the per-kernel mean
of each counter.  Usage: rocprof_summary.py <results.db> [...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("## %s" % path)
        try:
            rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        except sqlite3.Error:
            rows = []
        if rows:
            print("%-78s %6s %12s %12s %7s" % ("kernel", "calls", "avg_us", "total_us", "%"))
            for name, calls, total, avg, pct in rows[:12]:
                print("%-78s %6d %12.2f %12.1f %7.2f" % (name[:78], calls, float(avg), float(total), float(pct)))
        try:
            rows = list(cur.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                "from counters_collection group by kernel_name, counter_name order by avg(value) desc"))
        except sqlite3.Error:
            rows = []
        if rows:
            print("%-70s %-12s %6s %14s %14s %14s %10s" % ("kernel", "counter", "n", "mean", "min", "max", "avg_us"))
            for k, c, n, a, mn, mx, du in rows[:16]:
                print("%-70s %-12s %6d %14.2f %14.2f %14.2f %10.1f" % (k[:70], c, n, a, mn, mx, (du or 0) / 1e3))
        print()


if __name__ == "__main__":
    main()
