import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Timing of the batched feature-table operations (W tables per launch) on the device vs the CPU oracle (1 core).
Output goes to profiles/r1_ftab_ops.txt (see profiles/README.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
import ftab_model as fm
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth

W, NEW = int(os.environ.get("W", "256")), 150
be = gf.Backend(device=0)
orc = oracle_lib.load()
rng = np.random.default_rng(1)
tabs = {"device": abi.FeatureTables(be.lib, "gfbe_", be.ctx, W, 4096), "oracle": abi.FeatureTables(orc.lib, "gfo_", None, W, 4096)}
state = dict(next_id=0, alive=[])
frames = []
for fc in range(12):
    ids, obs, state["alive"], state["next_id"] = fm.random_frame(rng, state["next_id"], state["alive"], NEW, p_lost=0.1)
    frames.append((ids, obs))
poses = np.tile(abi.pose_rows(synth.Scenario(seed=1, n_landmarks=0).truth_state(0)["pose"]).ravel(), (W, 1))
tic_ric = np.tile(np.concatenate([np.zeros(3), np.eye(3).ravel()]), (W, 1))
res = {}
for name, T in tabs.items():
    t = {}
    for fc in range(11):
        ids, obs = frames[fc]
        t0 = time.perf_counter()
        T.add_frame([fc] * W, [ids] * W, [obs] * W, [0.0] * W)
        t["add_frame (frame %d)" % fc] = time.perf_counter() - t0
    n = T.size()
    t0 = time.perf_counter(); T.triangulate(poses, tic_ric); t["triangulate"] = time.perf_counter() - t0
    L = int((T.download(0)["n_obs"] >= 4).sum())
    x = [1.0 / rng.uniform(0.5, 9, L)] * W
    t0 = time.perf_counter(); T.set_depth(x); t["set_depth"] = time.perf_counter() - t0
    t0 = time.perf_counter(); T.check_outliers(poses, tic_ric, 1); t["movingConsistencyCheckW (first call)"] = time.perf_counter() - t0
    t0 = time.perf_counter(); T.check_outliers(poses, tic_ric, 1); t["movingConsistencyCheckW"] = time.perf_counter() - t0
    PR = np.tile(np.concatenate([np.zeros(3), np.eye(3).ravel()]), (W, 1))
    t0 = time.perf_counter(); T.remove_back_shift_depth(PR, PR); t["remove_back_shift_depth"] = time.perf_counter() - t0
    ids, obs = frames[11]
    t0 = time.perf_counter(); T.add_frame([10] * W, [ids] * W, [obs] * W, [0.0] * W); t["add_frame (steady state)"] = time.perf_counter() - t0
    t0 = time.perf_counter(); T.remove_front([10] * W); t["remove_front"] = time.perf_counter() - t0
    t0 = time.perf_counter(); T.remove_failures(); t["remove_failures"] = time.perf_counter() - t0
    res[name] = (t, int(n[0]))
print("feature-table operations, W = %d tables per call, ~%d features per table (host-call wall time incl. argument staging)" % (W, res["device"][1]))
print("%-32s %12s %12s %8s" % ("operation", "device ms", "oracle ms", "ratio"))
for k in res["device"][0]:
    if k.startswith("add_frame (frame") and not k.endswith("10)"):
        continue
    a, b = res["device"][0][k] * 1e3, res["oracle"][0][k] * 1e3
    print("%-32s %12.3f %12.3f %8.1f" % (k, a, b, b / a))
