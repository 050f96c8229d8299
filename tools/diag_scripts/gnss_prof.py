import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch  # noqa
import numpy as np
from _gfbe_import import gf
import gnss_window_cases as gw
abi = gf.abi
be = gf.Backend(device=0)
scn, tru, snap = gw.gnss_window(seed=81, L=150, n_per_frame=8)
plain = dict(snap); plain.pop("gnss")
for name, s in (("gnss", snap), ("plain", plain)):
    for _ in range(3): be.solve(s, abi.MARGIN_OLD)
    t0 = time.time()
    for _ in range(10): r = be.solve(s, abi.MARGIN_OLD)
    print(name, "ms per solve", (time.time() - t0) * 100, "iters", r["summary"]["iterations"], r["perf"])
    be.profile_enable(True); be.profile_reset()
    for _ in range(5): be.solve(s, abi.MARGIN_OLD)
    prof = be.profile()
    be.profile_enable(False)
    for p in sorted(prof, key=lambda p: -p["total_ms"]):
        print("   %-22s launches %4d  total %8.3f ms  avg %8.2f us" % (p["name"], p["launches"], p["total_ms"], 1e3 * p["total_ms"] / max(p["launches"], 1)))
