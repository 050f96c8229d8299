import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch  # noqa
import numpy as np
from _gfbe_import import gf
import gnss_window_cases as gw
abi = gf.abi
o = abi.default_options(); o.max_num_iterations = 1
be = gf.Backend(device=0, options=o)
scn, tru, snap = gw.gnss_window(seed=81, L=150, n_per_frame=8)
for rep in range(3):
    b = be.batch_upload([snap])
    try:
        b.solve(abi.MARGIN_OLD)
        r = b.download()
    except Exception as e:
        print("solve failed (ablation build?)", str(e)[:60])
    ts = b.debug_timing(0); tg = b.debug_timing(1)
    print("k_solve_big phases us:", [round((ts[i + 1] - ts[i]) * 0.01, 1) for i in range(5)], "total", round((ts[5] - ts[0]) * 0.01, 1))
    print("panel 10 us: k-loop", round((ts[9] - ts[8]) * 0.01, 2), "diag", round((ts[10] - ts[9]) * 0.01, 2), "barrier", round((ts[11] - ts[10]) * 0.01, 2), "LW^T+store", round((ts[12] - ts[11]) * 0.01, 2), "barrier", round((ts[13] - ts[12]) * 0.01, 2))
    print("k_gnss mode0 us: eval", round((tg[1] - tg[0]) * 0.01, 1), "cells", round((tg[3] - tg[1]) * 0.01, 1), "owners", round((tg[2] - tg[3]) * 0.01, 1), " mode2:", [round((tg[16 + i + 1] - tg[16 + i]) * 0.01, 1) for i in range(1)])
    b.free()
