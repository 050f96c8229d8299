import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Diagnostic (not a test): prints GPU-vs-oracle deviations for several windows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _gfbe_import import gf
import oracle_lib
abi, synth = gf.abi, gf.synth
be = gf.Backend(0); orc = oracle_lib.load()

def prior_window(seed, L):
    scn = synth.Scenario(seed=seed, n_landmarks=L, use_wheel=True)
    rA = orc.solve(scn.window(0), abi.MARGIN_OLD)
    return scn.window(1, state=synth.shift_state_for_next_window(scn, rA["state"], 1), prior=rA["prior"])

cases = {"cfg1": (synth.Scenario(seed=20250708, n_landmarks=200, use_wheel=False).window(0), abi.MARGIN_NONE),
         "cfg2": (prior_window(20250709, 2000), abi.MARGIN_OLD),
         "p400": (prior_window(53, 400), abi.MARGIN_SECOND_NEW),
         "w300": (synth.Scenario(seed=54, n_landmarks=300).window(0), abi.MARGIN_OLD)}
for name, (snap, flag) in cases.items():
    w, g = orc.solve(snap, flag), be.solve(snap, flag)
    sw, sg = w["summary"], g["summary"]
    ch = np.abs(np.array(sg["cost_history"]) - np.array(sw["cost_history"])) / np.array(sw["cost_history"])
    ate = np.sqrt(((g["state"]["pose"][:, :3] - w["state"]["pose"][:, :3]) ** 2).sum(axis=1).mean())
    rot = max(2 * np.linalg.norm(synth.qmul(synth.qinv(w["state"]["pose"][i, 3:]), g["state"]["pose"][i, 3:])[:3]) for i in range(11))
    print(name, "acc", sg["accepted"] == sw["accepted"], "it", sg["iterations"], sw["iterations"], "cost_hist_rel", ch.max(),
          "final_rel", abs(sg["final_cost"] - sw["final_cost"]) / sw["final_cost"], "ATE", ate, "rot", rot,
          "sb", np.abs(g["state"]["speed_bias"] - w["state"]["speed_bias"]).max(),
          "exw", np.abs(g["state"]["ex_pose_wheel"] - w["state"]["ex_pose_wheel"]).max(),
          "feat_rel", (np.abs(g["feature"] - w["feature"]) / np.abs(w["feature"])).max())
    if w["prior"] is not None:
        pw, pg = w["prior"], g["prior"]
        Aw, Ag = pw["J0"].T @ pw["J0"], pg["J0"].T @ pg["J0"]
        bw, bg = pw["J0"].T @ pw["r0"], pg["J0"].T @ pg["r0"]
        print("   prior n", pg["n"], pw["n"], "ids", pg["block_id"].tolist() == pw["block_id"].tolist(), "A rel", np.abs(Ag - Aw).max() / np.abs(Aw).max(),
              "b rel", np.abs(bg - bw).max() / max(1, np.abs(bw).max()), "x0", np.abs(pg["x0"] - pw["x0"]).max(),
              "r0norm", np.linalg.norm(pg["r0"]), np.linalg.norm(pw["r0"]))
