#!/usr/bin/env python3
"""Print the last N kernel dispatches of a rocprofv3 rocpd database in time order: start offset,
duration and the idle gap before each (all in us) -- the per-level timeline of one evaluation."""
import sqlite3
import sys


def main(db, n):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    rows = rows[-n:] if n > 0 else rows[-1:]
    t0 = rows[0][1]
    prev_end = rows[0][1]
    for name, st, en, gx, wx in rows:
        short = name.split("(")[0].replace("void mbamd::", "").replace("mbamd::", "")[:44]
        print("%9.1f  dur %8.1f  gap %6.1f  wgs %7d  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3,
                                                          gx // max(1, wx), short))
        prev_end = en
    print("span %.1f us" % ((rows[-1][2] - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
