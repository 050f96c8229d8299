import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Workload for a kernel trace of ONE resident 2k-landmark window re-solved N times (the counterpart of h2h_trace.py: the same window, no upload
between the solves) — per-kernel durations of the two side by side show what a fresh upload costs the kernels that follow it."""
import os, time
import numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
one = be.batch_upload([snap])
ts = []
for _ in range(int(os.environ.get("N", "40"))):
    t0 = time.perf_counter(); one.solve(abi.MARGIN_OLD); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    time.sleep(0.002)
print("resident: median %.4f ms, p10 %.4f" % (np.median(ts[10:]) * 1e3, np.percentile(ts[10:], 10) * 1e3))
