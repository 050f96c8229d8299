import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Wall time of gfbe_solve_window from host buffers (upload + solve + marginalise + download + free), one window per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
for L in (200, 2000):
    scn = synth.Scenario(seed=5, n_landmarks=L, use_wheel=True)
    r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
    wh = abi.WindowHolder(snap)
    for _ in range(5):
        be.solve(wh, abi.MARGIN_OLD)
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        res = be.solve(wh, abi.MARGIN_OLD)
    dt = (time.perf_counter() - t0) / n
    b = be.batch_upload([wh]); b.solve(abi.MARGIN_OLD)
    b.download()
    t0 = time.perf_counter()
    for _ in range(n):
        b.solve(abi.MARGIN_OLD)
    b.download()            # gfbe_batch_solve only enqueues
    dr = (time.perf_counter() - t0) / n
    b.download()            # drain the queued solves
    tu = tf = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); b2 = be.batch_upload([wh]); t1 = time.perf_counter(); b2.free(); t2 = time.perf_counter()
        tu += t1 - t0; tf += t2 - t1
    du = (tu + tf) / n
    print("  upload %.3f ms, free %.3f ms" % (tu / n * 1e3, tf / n * 1e3))
    t0 = time.perf_counter()
    for _ in range(n):
        b.download()
    dd = (time.perf_counter() - t0) / n
    print("L=%d: gfbe_solve_window from host buffers %.3f ms | resident solve %.3f ms | upload+free %.3f ms | download %.3f ms | iterations %d"
          % (L, dt * 1e3, dr * 1e3, du * 1e3, dd * 1e3, res["summary"]["iterations"]))
