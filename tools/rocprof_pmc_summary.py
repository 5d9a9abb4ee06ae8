#!/usr/bin/env python
"""Per-kernel PMC counter summary from a rocprofv3 (rocpd / SQLite) database collected with --kernel-trace --pmc ...
usage: tools/rocprof_pmc_summary.py <results.db> [out.txt] [--dispatches REGEX]   (prints avg per dispatch and dispatch count;
--dispatches: additionally one line "D <dispatch_id> <counter> <value> <grid_x> <kernel>" per dispatch whose kernel name matches REGEX, in
dispatch order -- for launches of ONE kernel that differ by shape, tools/conv_layers_pmc.py)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    rx = None
    if "--dispatches" in sys.argv:
        i = sys.argv.index("--dispatches")
        rx = sys.argv[i + 1]
        del sys.argv[i:i + 2]
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
    if rx and dcol:
        import re
        pat = re.compile(rx)
        gcol = "grid_size_x" if "grid_size_x" in cols else ("grid_size" if "grid_size" in cols else None)
        q = "select %s, %s, %s, sum(%s), %s from counters_collection group by %s, %s, %s order by %s" % (
            dcol, kcol, ccol, vcol, ("max(%s)" % gcol) if gcol else "0", dcol, kcol, ccol, dcol)
        for d, k, c, v, g in cur.execute(q):
            if pat.search(str(k)):
                out.write("D %d %s %.6g %d %s\n" % (d, c, v, g or 0, str(k)[:100]))


if __name__ == "__main__":
    main()
