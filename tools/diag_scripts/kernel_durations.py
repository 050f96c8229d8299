"""Sorted durations of the kernels whose name contains one of the given substrings, from a rocprofv3 rocpd database, in launch
order: python kernel_durations.py results.db k_dense_tp k_dense_raw"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute("select %s, start, end from kernels order by start" % name_col))
for key in sys.argv[2:]:
    d = [(e - s) / 1e3 for n, s, e in rows if key in n]
    tail = d[-48:]
    print(key, "launches", len(d), "last 48 in launch order (us):", " ".join("%.0f" % v for v in tail))
