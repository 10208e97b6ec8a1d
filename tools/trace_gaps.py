#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run (rocpd sqlite): start, duration and the gap to the previous kernel.
python tools/trace_gaps.py <dir-or-db> [--last N]"""
import glob, os, sqlite3, sys
path = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 40
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
for db in dbs:
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-last:]
    prev = None
    t0 = rows[0][1]
    for name, s, e in rows:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print(f"{(s - t0) / 1e3:10.1f} us  +{gap:8.1f} gap  {(e - s) / 1e3:9.1f} us  {name[:70]}")
        prev = e
