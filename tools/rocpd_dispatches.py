#!/usr/bin/env python3
"""Per-dispatch listing of a rocprofv3 rocpd database (kernel-trace): start (us from the first dispatch), duration, grid, kernel name --
the last N dispatches whose name contains a pattern.    python tools/rocpd_dispatches.py x.db [pattern] [N]"""
import sqlite3
import sys


def main(db, pat="", n=80):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    gx = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols]
    gy = [c for c in ("grid_y", "grid_size_y") if c in cols]
    gz = [c for c in ("grid_z", "grid_size_z") if c in cols]
    sel = "start, end, duration, name" + "".join(", " + c[0] for c in (gx, gy, gz) if c)
    rows = con.execute("select %s from kernels order by start" % sel).fetchall()
    t0 = rows[0][0] if rows else 0
    rows = [r for r in rows if pat in r[3]][-n:]
    prev_end = None
    for r in rows:
        gap = (r[0] - prev_end) / 1e3 if prev_end is not None else 0.0
        print("%12.1f us  gap %7.1f  dur %8.1f us  grid %-18s %s" % ((r[0] - t0) / 1e3, gap, r[2] / 1e3, "x".join(str(v) for v in r[4:]), r[3][:70]))
        prev_end = r[1]


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 80)
