#!/usr/bin/env python
"""A stretch of the kernel timeline of a rocprofv3 rocpd database, per stream / queue: which kernel of which part runs when (round 5:
what the candidate pass of a throughput batch waits for). usage: timeline_rocpd.py results.db out.txt [t0_ms] [len_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = [c for c in cols if "stream" in c] or [c for c in cols if "queue" in c]
qcol = qcol[0]
rows = list(cur.execute("select %s, start, end, %s from kernels order by start" % (name_col, qcol)))
try:      # memory copies of the same run (rocprofv3 --memory-copy-trace), if the database has them
    mcols = [r[1] for r in cur.execute("pragma table_info(memory_copies)")]
    if mcols:
        print("memory_copies columns:", mcols)
        sz = "size" if "size" in mcols else ([c for c in mcols if "size" in c or "bytes" in c] or ["0"])[0]
        nm = "name" if "name" in mcols else "'copy'"
        rows += [("%s %s B" % (n, b), s0, e0, "copy") for n, b, s0, e0 in cur.execute("select %s, %s, start, end from memory_copies" % (nm, sz))]
        rows.sort(key=lambda r: r[1])
except Exception as ex:
    print("(no memory copies: %r)" % (ex,))
t_first = rows[0][1]
t0 = t_first + int(float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else rows[len(rows) // 2][1]
ln = int(float(sys.argv[4]) * 1e6) if len(sys.argv) > 4 else 6_000_000
out = []
for name, s, e, q in rows:
    if s >= t0 and s < t0 + ln:
        short = name.split("(")[0].replace("void ", "").replace("gfd::", "")[-44:]
        out.append("%9.1f %9.1f %8.1f  q=%-6s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))
open(sys.argv[2], "w").write("start_us end_us dur_us stream kernel\n" + "\n".join(out) + "\n")
print(len(out), "dispatches written")
