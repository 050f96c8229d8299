import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(device=0)
scn = synth.Scenario(seed=7, n_landmarks=200, use_wheel=True)
snap = scn.window(0)
for i in range(4):
    r = be.solve(snap, abi.MARGIN_OLD)
    print(i, r["summary"]["final_cost"], r["prior"]["n"] if r["prior"] else None, flush=True)
ev = be.eval_factors(snap, robustify=True)
print("eval", ev["cost"], flush=True)
r = be.solve(snap, abi.MARGIN_OLD); print("after eval", r["summary"]["final_cost"], flush=True)
