#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 (rocpd / sqlite) outputs: python tools/pmc_dump.py <dir-or-db> ... [--kernel substr]"""
import glob, os, sqlite3, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = None
if "--kernel" in sys.argv:
    want = sys.argv[sys.argv.index("--kernel") + 1]
    args = [a for a in args if a != want]
for path in args:
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for db in dbs:
        con = sqlite3.connect(db)
        cols = [c[1] for c in con.execute("pragma table_info(counters_collection)")]
        q = "select name, counter_name, avg(counter_value), count(*), avg(duration) from pmc_events group by name, counter_name"
        rows = con.execute(q).fetchall()
        for name, cn, v, n, dur in rows:
            if want and want not in name:
                continue
            print(f"{name[:60]:60s} {cn:28s} {v:16.1f}  n={n} dur_us={dur/1e3 if dur else 0:.1f}")
