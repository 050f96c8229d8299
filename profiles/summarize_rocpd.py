#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table
(calls, total / average / min / max duration in microseconds, share of GPU time)."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute("select %s, start, end from kernels" % name_col))
    stats = {}
    for name, s, e in rows:
        d = (e - s) / 1e3
        st = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    tot = sum(v[1] for v in stats.values())
    lines = ["%-70s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        short = name.replace("(anonymous namespace)::", "").split("(")[0][-70:]
        lines.append("%-70s %8d %14.1f %12.2f %12.2f %12.2f %6.2f%%" % (short, v[0], v[1], v[1] / v[0], v[2], v[3], 100 * v[1] / tot))
    text = "\n".join(lines)
    print(text)
    if out_path:
        open(out_path, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
