import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch  # noqa
import numpy as np
import oracle_lib
from _gfbe_import import gf
import gnss_window_cases as gw
abi = gf.abi
o = oracle_lib.load()
be = gf.Backend(device=0)
for seed, npf in ((81, 8), (85, 3)):
    scn, tru, snap = gw.gnss_window(seed=seed, L=150, n_per_frame=npf)
    for flag in (abi.MARGIN_OLD,):
        want = o.solve(snap, flag)
        t0 = time.time(); got = be.solve(snap, flag); t1 = time.time()
        sw, sg = want["summary"], got["summary"]
        print("seed", seed, "status", got["status"], "iters", sg["iterations"], sw["iterations"], "acc", sg["accepted"], sw["accepted"], "term", sg["termination"], sw["termination"])
        print(" cost_hist want", np.array(sw["cost_history"]))
        print(" cost_hist got ", np.array(sg["cost_history"]))
        n = min(len(sw["cost_history"]), len(sg["cost_history"]))
        print(" rel cost dev", np.abs(np.array(sg["cost_history"][:n]) / np.array(sw["cost_history"][:n]) - 1).max(), "final", abs(sg["final_cost"] / sw["final_cost"] - 1))
        print(" pose dev", np.abs(got["state"]["pose"] - want["state"]["pose"]).max(), "sb dev", np.abs(got["state"]["speed_bias"] - want["state"]["speed_bias"]).max())
        a, b = want["state"]["gnss_state"], got["state"]["gnss_state"]
        print(" dt dev", np.abs(a["rcv_dt"] - b["rcv_dt"]).max(), "ddt dev", np.abs(a["rcv_ddt"] - b["rcv_ddt"]).max(), "anc dev", np.abs(a["anc_ecef"] - b["anc_ecef"]).max(), "yaw", a["yaw_enu_local"], b["yaw_enu_local"])
        print(" feature rel dev", np.abs(got["feature"] / want["feature"] - 1).max())
        pw, pg = want["prior"], got["prior"]
        if pw is not None and pg is not None:
            print(" prior ids equal", pw["block_id"].tolist() == pg["block_id"].tolist(), "n", pw["n"], pg["n"])
            if pw["n"] == pg["n"]:
                Aw, Ag = pw["J0"].T @ pw["J0"], pg["J0"].T @ pg["J0"]
                print(" prior A rel dev", np.abs(Ag - Aw).max() / np.abs(Aw).max(), "b dev", np.abs(pg["J0"].T @ pg["r0"] - pw["J0"].T @ pw["r0"]).max())
        print(" perf", got["perf"], "wall", t1 - t0)
        for _ in range(3):
            t0 = time.time(); be.solve(snap, flag); print("  wall ms", (time.time() - t0) * 1e3)
