import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""BASELINE configs[4] shape: the cfg-2 window (2k landmarks, wheel, prior) + 2000 LiDAR point-to-plane factors on the newest
pose. Resident-batch throughput and single-window time with and without the scan; CPU oracle on one core beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import oracle_lib
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
orc = oracle_lib.load()
scn = synth.Scenario(seed=20250712, n_landmarks=2000, use_wheel=True)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
joint = dict(snap, lio=synth.lidar_block(scn, 1, n=2000, seed=3, outliers=0.05))
for name, s in (("VIO window", snap), ("VIO + 2000 LiDAR factors", joint)):
    wh = abi.WindowHolder(s)
    out = []
    for B in (1, 512):
        b = be.batch_upload([wh] * B)
        b.solve(abi.MARGIN_OLD); torch.cuda.synchronize()
        n = 20 if B == 1 else 5
        t0 = time.perf_counter()
        for _ in range(n):
            b.solve(abi.MARGIN_OLD)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out.append((B, dt))
        b.free()
    t0 = time.perf_counter(); orc.solve(wh, abi.MARGIN_OLD); tc = time.perf_counter() - t0
    print("%-28s B=1: %.3f ms/solve | B=512: %.2f ms/step = %.0f solves/s | CPU oracle 1 core: %.1f ms/solve"
          % (name, out[0][1] * 1e3, out[1][1] * 1e3, 512 / out[1][1], tc * 1e3))
