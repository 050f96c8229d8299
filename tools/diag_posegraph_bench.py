import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Timing of the BASELINE configs[3] pose graph (5 000 poses, 5 Levenberg-Marquardt iterations): device vs CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be, orc = gf.Backend(0), oracle_lib.load()
for n in (5000, 50000):
    g = synth.pose_graph(n=n) if n == 5000 else None
    if g is None:
        base = synth.pose_graph(n=5000)
        reps = n // 5000
        # a longer chain with the same statistics: tile the relative measurements, rebuild the dead-reckoned guess
        rel_meas = np.tile(base["rel_meas"], (reps, 1))[: n - 1]
        rel_meas = np.vstack([rel_meas, base["rel_meas"][: n - 1 - len(rel_meas)]]) if len(rel_meas) < n - 1 else rel_meas
        pose = np.zeros((n, 7)); pose[0] = base["pose"][0]
        pgo = abi.PoseGraph(orc.lib, "gfo_", None)
        for k in range(n - 1):
            pose[k + 1] = pose[k]   # placeholder; the solver only needs a consistent starting point
        pose[:, :3] = np.cumsum(np.vstack([np.zeros(3), np.tile(np.diff(base["pose"][:, :3], axis=0), (reps + 1, 1))[: n - 1]]), axis=0)
        pose[:, 3:] = np.tile(base["pose"][:, 3:], (reps + 1, 1))[:n]
        fix_i = np.arange(0, n, 10, dtype=np.int32)
        g = dict(pose=pose, rel_i=np.arange(n - 1, dtype=np.int32), rel_meas=rel_meas, fix_i=fix_i,
                 fix_meas=np.column_stack([pose[fix_i, :3] + np.random.default_rng(1).normal(0, 0.5, (len(fix_i), 3)), np.full(len(fix_i), 0.5)]))
    dev, ref = abi.PoseGraph(be.lib, "gfbe_", be.ctx), abi.PoseGraph(orc.lib, "gfo_", None)
    dev.solve(g)
    tds = []
    for _ in range(5):
        t0 = time.perf_counter(); rd = dev.solve(g); tds.append(time.perf_counter() - t0)
    td = sorted(tds)[2]      # median of five calls
    t0 = time.perf_counter(); rr = ref.solve(g); tr = time.perf_counter() - t0
    print("n = %6d poses: device %.2f ms, oracle (1 core) %.2f ms, ratio %.1f; iterations %d / %d, max |dp| %.1e m" %
          (n, td * 1e3, tr * 1e3, tr / td, rd["summary"]["iterations"], rr["summary"]["iterations"], np.abs(rd["pose"][:, :3] - rr["pose"][:, :3]).max()))
