#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) as the --stats table: per kernel calls,
total / average / min / max duration (ns) and share of GPU kernel time."""
import sqlite3
import sys


def main(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-100s %8s %14s %12s %10s %10s %7s" % ("Name", "Calls", "TotalNs", "AverageNs", "MinNs", "MaxNs", "Pct"))
    for name, n, tot, avg, mn, mx in rows:
        print("%-100s %8d %14d %12.0f %10d %10d %7.2f" % (name[:100], n, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
