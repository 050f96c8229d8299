import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Long-horizon check of the stream harness: 100 keyframes (90 consecutive optimization() calls, each fed by the previous one's
state, prior and depths) on the HIP library (tables on the device, device hand-over) and on the CPU oracle."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, stream = gf.abi, gf.stream
NKF = int(os.environ.get("NKF", "100"))
S = stream.Stream(seed=11, n_kf=NKF, new_per_frame=40)
opts = dict(min_parallax=14.0 / 600, depth_threshold=6.0)
be, orc = gf.Backend(0), oracle_lib.load()
To = abi.FeatureTables(orc.lib, "gfo_", None, 1, 8192, options=opts)
Tg = abi.FeatureTables(be.lib, "gfbe_", be.ctx, 1, 8192, options=opts)
t0 = time.time(); ref = stream.run_stream(orc, To, S, lambda st, flag: orc.lib.gfo_slide_window_state(C.byref(st), int(flag))); tr = time.time() - t0
t0 = time.time(); got = stream.run_stream(be, Tg, S, lambda st, flag: be.lib.gfbe_slide_window_state(C.byref(st), int(flag)), device_handoff=True); tg = time.time() - t0
n = len(ref["traj"])
dp = np.linalg.norm(got["traj"][:, :3] - ref["traj"][:, :3], axis=1)
err = np.array([np.linalg.norm(got["traj"][i, :3] - S.truth_pose(10 + i)[0]) for i in range(n)])
print("%d consecutive solves: oracle %.1f s, device %.2f s; flags equal: %s (%d x MARGIN_OLD, %d x SECOND_NEW); iterations equal: %s; table sizes equal: %s"
      % (n, tr, tg, got["flags"] == ref["flags"], sum(f == abi.MARGIN_OLD for f in ref["flags"]), sum(f == abi.MARGIN_SECOND_NEW for f in ref["flags"]),
         got["iterations"] == ref["iterations"], got["n_features"] == ref["n_features"]))
print("device vs oracle position: max %.2e m (first 10: %.1e, last 10: %.1e); error against the ground truth: max %.3f m, final %.3f m"
      % (dp.max(), dp[:10].max(), dp[-10:].max(), err.max(), err[-1]))
