#!/usr/bin/env python
"""Per-kernel PMC counter summary from a rocprofv3 (rocpd / SQLite) database collected with --kernel-trace --pmc ...
usage: tools/rocprof_pmc_summary.py <results.db> [out.txt]    (prints avg per dispatch and dispatch count)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "a") if len(sys.argv) > 2 else sys.stdout
    cur = sqlite3.connect(db).cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    out.write("# PMC summary of %s\n" % db)
    if "counters_collection" not in views:
        out.write("# no counters_collection view; views: %s\n" % views)
        return
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    out.write("# columns: %s\n" % cols)
    kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    ccol = "counter_name" if "counter_name" in cols else "pmc_name"
    vcol = "value" if "value" in cols else "counter_value"
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    if dcol:
        q = ("select k, c, count(*), avg(v), min(v), max(v) from (select %s as k, %s as c, %s as d, sum(%s) as v from counters_collection "
             "group by %s, %s, %s) group by k, c order by k, c" % (kcol, ccol, dcol, vcol, kcol, ccol, dcol))
    else:
        q = "select %s, %s, count(*), avg(%s), min(%s), max(%s) from counters_collection group by 1, 2 order by 1, 2" % (kcol, ccol, vcol, vcol, vcol)
    out.write("%-60s %-32s %8s %16s %16s %16s\n" % ("kernel", "counter", "launches", "avg/launch", "min", "max"))
    for r in cur.execute(q):
        out.write("%-60s %-32s %8d %16.6g %16.6g %16.6g\n" % (str(r[0])[:60], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
