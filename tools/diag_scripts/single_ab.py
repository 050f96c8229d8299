import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""One window at a time (bench.py's default 2000-landmark window with prior, and BASELINE configs[0]'s 200-landmark window) under several
builds of the library, one box: resident re-solve and gfbe_solve_window host to host, medians; every output compared bit for bit with the
first build's. usage: single_ab.py default <other.so> ..."""
import sys, time
import numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
be0 = gf.Backend(0)
r = be0.solve(scn.window(0), abi.MARGIN_OLD)
big = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
be0.close()
cfg1 = synth.Scenario(seed=7, n_landmarks=200, use_wheel=False).window(0)
ref = {}
for var in sys.argv[1:]:
    be = gf.Backend(0) if var == "default" else gf.Backend(0, so=var)
    line = var[-28:]
    for name, snap in (("2k", big), ("cfg1", cfg1)):
        one = be.batch_upload([snap])
        for _ in range(10): one.solve(abi.MARGIN_OLD)
        torch.cuda.synchronize()
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); one.solve(abi.MARGIN_OLD); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res = one.download()[0]; one.free()
        h = abi.WindowHolder(snap); th = []
        for _ in range(110):
            t0 = time.perf_counter(); be.solve_raw(h, abi.MARGIN_OLD); th.append(time.perf_counter() - t0)
        key = (res["summary"]["final_cost"], res["state"]["pose"].tobytes(), res["feature"].tobytes(), res["prior"]["J0"].tobytes() if res.get("prior") else b"")
        same = ref.setdefault(name, key) == key
        line += "  %s: resident %.4f ms, host to host %.4f ms%s" % (name, np.median(ts) * 1e3, np.median(th[10:]) * 1e3, "" if same else " (RESULT DIFFERS)")
    print(line, flush=True)
    be.close()
