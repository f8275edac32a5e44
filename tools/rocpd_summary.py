#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace/--stats or --pmc run) as text:
per-kernel calls / total / average / min / max duration, and per-kernel mean counter values.
Usage: python tools/rocpd_summary.py <results.db> [> profiles/xxx.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print("# rocprofv3 summary of", path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "min(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-90s %7s %12s %12s %12s %12s %6s %8s %5s %5s %5s %7s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid_x", "wg_x", "vgpr", "sgpr", "lds"))
    for r in rows:
        name = r[0] if len(r[0]) <= 90 else r[0][:87] + "..."
        print("%-90s %7d %12.1f %12.3f %12.3f %12.3f %6.2f %8d %5d %5d %5d %7d" % (
            name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7],
            (r[8] or 0) + (r[9] or 0), r[10] or 0, r[11] or 0))
    try:
        pm = c.execute(
            "select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value) "
            "from pmc_events group by name, counter_name order by avg(counter_value) desc").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print("\n# counters (per dispatch)")
        print("%-90s %-16s %7s %16s %16s %16s" % ("kernel", "counter", "calls", "mean", "min", "max"))
        for r in pm:
            name = r[0] if len(r[0]) <= 90 else r[0][:87] + "..."
            print("%-90s %-16s %7d %16.1f %16.1f %16.1f" % (name, r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main(sys.argv[1])
