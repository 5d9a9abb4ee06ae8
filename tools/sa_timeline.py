#!/usr/bin/env python
"""Every set-abstraction dispatch (register / LDS / f64 re-evaluation kernels) of the LAST encoder pass in a rocprofv3 --kernel-trace
database, in launch order: start offset, duration, gap to the previous one.   usage: tools/sa_timeline.py <results.db>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
sa = [r for r in rows if "sa_" in r[0]]
firsts = [i for i, r in enumerate(sa) if "sa_small_kernel_w3<16, 16, 16, 32>" in r[0]]
last = sa[firsts[-1]:]
t0, prev = last[0][1], None
for n, s, e in last:
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, n[:60]))
    prev = e
