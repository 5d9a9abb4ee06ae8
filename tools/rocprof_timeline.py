#!/usr/bin/env python
"""Timeline of ONE bench step from a rocprofv3 --kernel-trace database: every dispatch between two consecutive CNF solves
(start offset, duration, queue, grid), to see what overlaps and where the main stream waits.
usage: tools/rocprof_timeline.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute("select name, start, end, %s, grid_x, grid_y, grid_z from kernels order by start" % q).fetchall()
    cnf = [i for i, r in enumerate(rows) if r[0].startswith("void cnf_rk4") or r[0].startswith("cnf_rk4")]
    # the accuracy guard (on by default since round 6) repeats every solve on the 64-point kernel: the step boundaries are the MAIN solves,
    # i.e. the launches of the kernel with the largest total time
    if cnf:
        tot = {}
        for i in cnf:
            tot[rows[i][0]] = tot.get(rows[i][0], 0) + rows[i][2] - rows[i][1]
        main_name = max(tot, key=tot.get)
        cnf = [i for i in cnf if rows[i][0] == main_name]
    if len(cnf) < 2:
        out.write("fewer than two CNF solves in the trace\n")
        return
    a, b = cnf[-2], cnf[-1]
    t0 = rows[a][2]
    out.write("# one step: dispatches after the end of CNF solve %d up to the end of solve %d (columns: kernel trace of %s: %s)\n" % (len(cnf) - 1, len(cnf), db, ",".join(cols)))
    out.write("%10s %10s %10s %6s  %-18s %s\n" % ("start_us", "dur_us", "end_us", "queue", "grid", "kernel"))
    for r in rows[a + 1:b + 1]:
        out.write("%10.1f %10.1f %10.1f %6s  %-18s %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3, r[3],
                                                         "%dx%dx%d" % (r[4], r[5], r[6]), r[0][:70]))


if __name__ == "__main__":
    main()
