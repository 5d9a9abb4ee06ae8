#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd / SQLite) result database into the per-kernel statistics table that
`rocprofv3 --kernel-trace --stats` describes (name, calls, total, average, min, max, share), plus the launch
geometry / register use of each kernel.   usage: tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(grid_x), max(grid_y), max(grid_z), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), "
                       "max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out.write("# rocprofv3 --kernel-trace --stats summary of %s\n" % db)
    out.write("# total kernel time %.3f ms over %d dispatches\n" % (tot / 1e6, sum(r[1] for r in rows)))
    out.write("%-64s %6s %12s %12s %12s %12s %6s  %-18s %5s %5s %5s %7s %7s\n" % (
        "kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "grid(max)", "wg", "vgpr", "agpr", "lds", "scratch"))
    for r in rows:
        out.write("%-64s %6d %12.3f %12.2f %12.2f %12.2f %6.2f  %-18s %5d %5d %5d %7d %7d\n" % (
            r[0][:64], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            "%dx%dx%d" % (r[6], r[7], r[8]), r[9], r[10], r[11], r[13], r[14]))


if __name__ == "__main__":
    main()
