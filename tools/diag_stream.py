import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, stream = gf.abi, gf.stream
orc = oracle_lib.load()
import os
RG = bool(int(os.environ.get('RGBD', '0')))
S = stream.Stream(seed=3, n_kf=28, new_per_frame=50, rgbd=RG)
opts = dict(min_parallax=14.0 / 600, depth_threshold=6.0)
To = abi.FeatureTables(orc.lib, "gfo_", None, 1, 8192, options=opts)
ref = stream.run_stream(orc, To, S, lambda st, flag: orc.lib.gfo_slide_window_state(C.byref(st), int(flag)), rgbd=RG)
for ms in (1, 0):
    o = abi.default_options(); o.marg_sqrt = ms
    be = gf.Backend(device=0, options=o)
    Tg = abi.FeatureTables(be.lib, "gfbe_", be.ctx, 1, 8192, options=opts)
    got = stream.run_stream(be, Tg, S, lambda st, flag: be.lib.gfbe_slide_window_state(C.byref(st), int(flag)), rgbd=RG)
    print("marg_sqrt", ms, "cost rel", np.abs(np.array(got["final_cost"]) / np.array(ref["final_cost"]) - 1).max(),
          "pos", np.abs(got["traj"][:, :3] - ref["traj"][:, :3]).max(), "quat", np.abs(got["traj"][:, 3:] - ref["traj"][:, 3:]).max())
    print(np.abs(got["traj"][:, :3] - ref["traj"][:, :3]).max(axis=1))
