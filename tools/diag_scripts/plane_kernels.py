import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""The plane + anchor window of tests/test_gpu_plane.py under the three factorisations of gfbe_options.solve_kernel, against the oracle:
how far the weakly observed plane roll / pitch quaternion ends from the oracle's with each order of elimination."""
import numpy as np
from _gfbe_import import gf
import oracle_lib
from plane_cases import plane_window
abi = gf.abi
orc = oracle_lib.load()
for anchor in (True, False):
    scn, snap = plane_window(anchor=anchor)
    want = orc.solve(snap, abi.MARGIN_OLD)
    for kernel in (1, 2, 3):
        o = abi.default_options(); o.solve_kernel = kernel
        be = gf.Backend(0, options=o)
        got = be.solve(snap, abi.MARGIN_OLD)
        print("anchor %d kernel %d: |plane_R - oracle| %.3e  plane_Z %.3e  final cost rel %.2e  iterations %d term %d" % (
            anchor, kernel, np.abs(got["state"]["plane_R"] - want["state"]["plane_R"]).max(), abs(got["state"]["plane_Z"] - want["state"]["plane_Z"]),
            abs(got["summary"]["final_cost"] - want["summary"]["final_cost"]) / want["summary"]["final_cost"], got["summary"]["iterations"], got["summary"]["termination"]))
        be.close()
