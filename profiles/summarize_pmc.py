#!/usr/bin/env python
"""Summarise the PMC values of a rocprofv3 rocpd database: per (kernel, grid) the mean over dispatches of each
counter (a counter's per-dispatch value = sum over its instances / dimensions)."""
import sqlite3
import sys
from collections import defaultdict


def table(cur, prefix):
    r = [n for (n,) in cur.execute("select name from sqlite_master where type='table'") if n.startswith(prefix)]
    return r[0] if r else None


def main(db_path, out_path=None, only=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tk = table(cur, "rocpd_kernel_dispatch_")
    ts = table(cur, "rocpd_info_kernel_symbol_")
    tp = table(cur, "rocpd_pmc_event_")
    ti = table(cur, "rocpd_info_pmc_")
    scols = [r[1] for r in cur.execute("pragma table_info(%s)" % ts)]
    ncol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    names = dict(cur.execute("select id, %s from %s" % (ncol, ts)))
    pmc = dict(cur.execute("select id, name from %s" % ti))
    disp = {}
    for eid, kid, gx, gy, gz, wx, s, e in cur.execute(
            "select event_id, kernel_id, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, start, end from %s" % tk):
        disp[eid] = (names.get(kid, str(kid)).split("(")[0], gx, gy, gz, wx, (e - s) / 1e3)
    per = defaultdict(lambda: defaultdict(float))      # event -> counter -> summed value
    for eid, pid, val in cur.execute("select event_id, pmc_id, value from %s" % tp):
        per[eid][pmc[pid]] += val
    agg = defaultdict(lambda: [0, 0.0, defaultdict(float)])
    for eid, cs in per.items():
        if eid not in disp:
            continue
        key = disp[eid][:5]
        a = agg[key]
        a[0] += 1
        a[1] += disp[eid][5]
        for k, v in cs.items():
            a[2][k] += v
    lines = []
    for key, (n, us, cs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if only and only not in key[0]:
            continue
        lines.append("%s grid=(%d,%d,%d) wg=%d dispatches=%d avg_us=%.2f" % (key[0][-40:], key[1], key[2], key[3], key[4], n, us / n))
        for k in sorted(cs):
            lines.append("    %-32s %18.1f per dispatch" % (k, cs[k] / n))
    text = "\n".join(lines)
    print(text)
    if out_path:
        open(out_path, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
