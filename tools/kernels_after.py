#!/usr/bin/env python3
"""Which kernels run AFTER the first dispatch whose name contains MARK, by count and time (rocprofv3 rocpd database):
shows a product path free of framework kernels once the workload has been generated.
    python tools/kernels_after.py results.db MARK"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
mark = sys.argv[2]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
first = next((i for i, r in enumerate(rows) if mark in r[0]), None)
if first is None:
    sys.exit("no dispatch of a kernel named *%s*" % mark)
before, after = rows[:first], rows[first:]
print("# %d dispatches before the first *%s* (workload generation, uploads), %d from it on" % (len(before), mark, len(after)))
for title, part in (("before", before), ("from the first *%s* on" % mark, after)):
    agg = {}
    for name, s, e in part:
        k = agg.setdefault(name, [0, 0])
        k[0] += 1
        k[1] += e - s
    fw = sum(v[0] for n, v in agg.items() if "at::native" in n or "rocclr" in n.lower())
    print("\n## %s: %d kernel names, %d dispatches of at::native / rocclr kernels" % (title, len(agg), fw))
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-100s %7d %12.1f us" % (name if len(name) <= 100 else name[:97] + "...", n, t / 1e3))
