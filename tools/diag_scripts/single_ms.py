import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""single-window latency + a checksum of the result, for the library GFBE_LIB selects (variants side by side)"""
import os, sys, time, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
be.set_stream(torch.cuda.current_stream().cuda_stream)
out = []
for L, wheel in ((2000, True), (200, False)):
    scn = synth.Scenario(seed=20250708 + 2, n_landmarks=L, use_wheel=wheel)
    r0 = be.solve(scn.window(0), abi.MARGIN_OLD)
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
        one = be.batch_upload([snap])
        for _ in range(5): one.solve(flag)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            t1 = time.perf_counter()
            for _ in range(20): one.solve(flag)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t1) / 20 * 1e3)
        res = one.download()[0]
        one.free()
        h = hashlib.sha1()
        for k, v in sorted(abi.flat_state(res["state"]).items()): h.update(np.ascontiguousarray(v).tobytes())
        h.update(np.ascontiguousarray(res["feature"]).tobytes()); h.update(np.ascontiguousarray(res["prior"]["J0"]).tobytes())
        s = res["summary"]
        out.append("L=%d flag=%d: %.4f ms (min of 5; median %.4f)  iters %d accepted %s final %.12e sha %s" % (L, flag, min(ts), sorted(ts)[2], s["iterations"], sum(s["accepted"]), s["final_cost"], h.hexdigest()[:12]))
# a batch of 7 different windows: still identical to the singles?
snaps = [synth.Scenario(seed=900 + i, n_landmarks=300 + 100 * i, use_wheel=bool(i & 1)).window(0) for i in range(7)]
singles = [be.solve(s, abi.MARGIN_OLD) for s in snaps]
batch = be.solve_batch(snaps, abi.MARGIN_OLD)
same = all(a["summary"] == b["summary"] and np.array_equal(a["state"]["pose"], b["state"]["pose"]) and np.array_equal(a["prior"]["J0"], b["prior"]["J0"]) for a, b in zip(singles, batch))
h = hashlib.sha1()
for a in singles: h.update(np.ascontiguousarray(a["state"]["pose"]).tobytes()); h.update(np.ascontiguousarray(a["prior"]["J0"]).tobytes())
out.append("7 singles == batch of 7: %s  sha %s" % (same, h.hexdigest()[:12]))
print("\n".join(out))
