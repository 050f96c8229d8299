import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""which not-cleared array of the slab is read before it is written? One array at a time filled with NaN patterns (GFBE_POISON_UNCLEARED=<name>)."""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from _gfbe_import import gf
import gnss_window_cases as gw
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=11, n_landmarks=300, use_wheel=True)
r0 = be.solve(scn.window(0), abi.MARGIN_OLD)
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
cases = {"single": [snap], "batch40": [snap if i % 2 else scn.window(0) for i in range(40)], "gnss": [gw.gnss_window(seed=81, L=150, n_per_frame=8)[2]],
         "empty": [synth.Scenario(seed=12, n_landmarks=0, use_wheel=True).window(0)]}
def run(snaps, flag):
    out = be.solve_batch(snaps, flag)
    return [(r["summary"]["final_cost"], r["summary"]["iterations"], float(np.abs(r["state"]["pose"]).sum()), None if r["prior"] is None else float(np.abs(r["prior"]["J0"]).sum())) for r in out]
names = ["prior_J0", "lm_obs", "lm_rec", "lm_hP", "vis_part", "mA", "mb", "mJ0", "mr0", "x", "xout", "pc", "prior_H", "H", "dbg_imu", "dbg_wheel", "dbg_prior", "rec", "mV",
         "vis_contrib", "solveY", "solveS", "gnss_J", "gnss_r", "gnss_cost", "gnss_marg", "dl_fix", "dl_feat", "dl_J0"]
os.environ.pop("GFBE_POISON_UNCLEARED", None)
ref = {(k, f): run(v, f) for k, v in cases.items() for f in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW)}
for n in names + ["1"]:
    os.environ["GFBE_POISON_UNCLEARED"] = "clean"
    for (k, f) in ref: run(cases[k], f)
    os.environ["GFBE_POISON_UNCLEARED"] = n
    bad = []
    for (k, f), want in ref.items():
        try:
            got = run(cases[k], f)
        except Exception as e:
            bad.append("%s/%d: %s" % (k, f, str(e)[:60])); continue
        if repr(got) != repr(want): bad.append("%s/%d" % (k, f))
    print("%-12s %s" % (n, "ok" if not bad else "DIFFERS: " + ", ".join(bad)), flush=True)
