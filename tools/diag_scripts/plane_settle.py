import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch  # noqa
import numpy as np
import oracle_lib
from _gfbe_import import gf
from plane_cases import plane_window, next_plane_window
abi = gf.abi
orc = oracle_lib.load()
scn, snap = plane_window(anchor=True)
want = orc.solve(snap, abi.MARGIN_OLD)
snap2 = next_plane_window(scn, snap, want)
for iters in (8, 12, 15):
    o = abi.default_options(); o.max_num_iterations = iters
    be = gf.Backend(device=0, options=o)
    w2 = orc.with_options(max_num_iterations=iters).solve(snap2, abi.MARGIN_OLD)
    g2 = be.solve(snap2, abi.MARGIN_OLD)
    sw, sg = w2["summary"], g2["summary"]
    print(iters, "iters", sw["iterations"], sg["iterations"], "term", sw["termination"], sg["termination"], "acc eq", sw["accepted"] == sg["accepted"],
          "final rel", abs(sg["final_cost"] / sw["final_cost"] - 1), "pose", np.abs(g2["state"]["pose"] - w2["state"]["pose"]).max(),
          "plane", np.abs(g2["state"]["plane_R"] - w2["state"]["plane_R"]).max(), "costs", np.array(sw["cost_history"])[-3:])
    be.close()
