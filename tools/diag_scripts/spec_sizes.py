import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Speculative linearisation on / off by window size: host-to-host time of one window of L landmarks (diagnostics build: GFBE_SPEC_TILES
overrides the tile limit of small batches)."""
import os, time
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
for L in [int(x) for x in os.environ.get("LS", "200,2000,3500,6000,10000").split(",")]:
    scn = synth.Scenario(seed=31 + L, n_landmarks=L, use_wheel=True)
    be0 = gf.Backend(0)
    r = be0.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
    be0.close()
    h = abi.WindowHolder(snap)
    out = []
    for spec in (0, 1, 0, 1):
        o = abi.default_options(); o.speculative_linearization = spec
        be = gf.Backend(0, options=o)
        ts = []
        for _ in range(120):
            t0 = time.perf_counter(); be.solve_raw(h, abi.MARGIN_OLD); ts.append(time.perf_counter() - t0)
        out.append(np.median(ts[20:]) * 1e3)
        be.close()
    print("L = %5d: host to host, speculative off %.4f / %.4f ms, on %.4f / %.4f ms" % (L, out[0], out[2], out[1], out[3]), flush=True)
