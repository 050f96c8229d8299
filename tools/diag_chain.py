import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Diagnostic: the chain-eliminated factorisation (gfbe_options.solve_kernel = 0, k_solve_chain) against the monolithic one
(solve_kernel = 1, k_solve) on the same windows: Gauss-Newton step of the first linearisation entry by entry, whole solves,
kernel times (profile API) for one window and under load."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth


def backend(kernel, iters=8):
    o = abi.default_options()
    o.solve_kernel = kernel
    o.max_num_iterations = iters
    return gf.Backend(0, options=o)


def windows(L=2000, seed=5):
    be = backend(1)
    scn = synth.Scenario(seed=seed, n_landmarks=L, use_wheel=True)
    w0 = scn.window(0)
    r = be.solve_batch([w0], abi.MARGIN_OLD)[0]
    w1 = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
    be.close()
    return w0, w1


w0, w1 = windows()
for name, snap in (("no prior", w0), ("with prior", w1)):
    ys = []
    for kern in (1, 0):
        be = backend(kern, iters=1)
        b = be.batch_upload([snap])
        b.solve(abi.MARGIN_NONE)
        ys.append((b.debug_vector(0), b.debug_vector(1), b.download()[0]["summary"]))
        b.free(); be.close()
    (y1, v1, s1), (y0, v0, s0) = ys
    print("%s: first GN step  max|y_mono| %.3e  max|y_chain - y_mono| %.3e  (v diff %.1e)  cost after 1 iter mono %.10e chain %.10e" % (
        name, np.abs(y1).max(), np.abs(y0 - y1).max(), np.abs(v0 - v1).max(), s1["final_cost"], s0["final_cost"]))
    bad = np.argsort(-np.abs(y0 - y1))[:6]
    print("   worst dims", [(int(a), float(y1[a]), float(y0[a])) for a in bad])
for name, snap in (("no prior", w0), ("with prior", w1)):
    res = []
    for kern in (1, 0):
        be = backend(kern)
        r = be.solve(snap, abi.MARGIN_OLD)
        res.append(r); be.close()
    a, b = res
    print("%s: full solve  iterations %d / %d  accepted %s / %s" % (name, a["summary"]["iterations"], b["summary"]["iterations"], a["summary"]["accepted"], b["summary"]["accepted"]))
    print("   final cost %.12e / %.12e   dpos %.2e  dsb %.2e" % (a["summary"]["final_cost"], b["summary"]["final_cost"],
          np.abs(a["state"]["pose"] - b["state"]["pose"]).max(), np.abs(a["state"]["speed_bias"] - b["state"]["speed_bias"]).max()))
# timing
for kern in (1, 0):
    be = backend(kern)
    for B in (1, 512):
        b = be.batch_upload([w1] * B)
        b.solve(abi.MARGIN_OLD); b.solve(abi.MARGIN_OLD)
        b.download()
        t0 = time.perf_counter()
        reps = 20 if B == 1 else 3
        for _ in range(reps):
            b.solve(abi.MARGIN_OLD)
        b.download()
        dt = (time.perf_counter() - t0) / reps
        be.profile_enable(True); be.profile_reset()
        b.solve(abi.MARGIN_OLD); b.download()
        prof = be.profile()
        be.profile_enable(False)
        ks = [p for p in prof if p["name"] == "k_solve"]
        print("kernel %d  B=%d: %.3f ms per batch solve; k_solve %.1f us per launch (%d launches)" % (kern, B, dt * 1e3, ks[0]["total_ms"] / ks[0]["launches"] * 1e3 if ks else -1, ks[0]["launches"] if ks else 0))
        if B == 1 and kern == 0:
            t = b.debug_timing(0)
            for i, nme in ((1, "prologue"), (2, "build")):
                print("   %-10s %7.2f us" % (nme, (t[i] - t[i - 1]) * 0.01))
            print("   %-10s %7.2f us" % ("pipeline", (t[15] - t[2]) * 0.01))
            print("   %-10s %7.2f us" % ("dense chol", (t[3] - t[15]) * 0.01))
            print("   %-10s %7.2f us" % ("backsub", (t[4] - t[3]) * 0.01))
            print("   %-10s %7.2f us" % ("gram", (t[5] - t[4]) * 0.01))
            print("   total %.2f us" % ((t[5] - t[0]) * 0.01))
            if os.environ.get("GFBE_LIB", "").endswith("chainstamp.so"):
                print("   build: loads issued %.2f | +chain loads %.2f | tile stores %.2f | chain stores %.2f | barrier %.2f" % tuple((t[i] - t[j]) * 0.01 for i, j in ((8, 1), (9, 8), (10, 9), (11, 10), (2, 11))))
                u = b.debug_timing(1)
                print("   chain wave: step starts (us from pipeline start):", " ".join("%.2f" % ((u[s] - t[2]) * 0.01) for s in range(0, 14)))
                print("   chain wave: work done at                        :", " ".join("%.2f" % ((u[14 + s] - t[2]) * 0.01) for s in range(0, 6)))
                print("   wide wave 2: work done at                       :", " ".join("%.2f" % ((u[20 + s] - t[2]) * 0.01) for s in range(0, 12)))
        b.free()
    be.close()
