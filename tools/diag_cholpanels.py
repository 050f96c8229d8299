import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Diagnostic (GFBE_CHOL_STAMP=1 builds only): time of every panel of the k_solve factorisation for one 2k-landmark window."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=5, n_landmarks=2000, use_wheel=True)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
b = be.batch_upload([snap])
b.solve(abi.MARGIN_OLD); b.solve(abi.MARGIN_OLD)
t0, t1 = b.debug_timing(0), b.debug_timing(1)
prev = t0[17]
print("panel: us (from the start of panel 0)")
for P in range(12):
    print("  P=%2d  %6.2f" % (P, (t1[P] - prev) * 0.01))
    prev = t1[P]
print("tile (0,0) before the loop: %.2f us; all: %.2f us" % ((t0[17] - t0[2]) * 0.01, (t0[3] - t0[2]) * 0.01))
