import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""The eigen square root of the new prior (marg_sqrt = 0) by divide & conquer (the library as built) against the QL iteration
(GFBE_QL_LIB: a build with -DGFBE_EIG_DC=0) and against the oracle, on windows of several shapes: information J0^T J0, J0^T r0, the
orthogonality of J0's rows (J0 J0^T must be diag(S)), |r0|^2, and the call's time."""
import os, time
import numpy as np
import torch
from _gfbe_import import gf
import oracle_lib
abi, synth = gf.abi, gf.synth
orc = oracle_lib.load()
ql = os.environ.get("GFBE_QL_LIB")
worst = dict(A_ql=0.0, b_ql=0.0, A_or=0.0, b_or=0.0, orth=0.0)
cases = [(20250708 + 2, 2000, True), (11, 300, True), (12, 150, False), (13, 900, True), (14, 60, True), (15, 3000, True), (16, 500, False)]
for seed, L, wheel in cases:
    scn = synth.Scenario(seed=seed, n_landmarks=L, use_wheel=wheel)
    o = abi.default_options(); o.marg_sqrt = 0
    be = gf.Backend(0, options=o)
    r = be.solve(scn.window(0), abi.MARGIN_OLD)
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
    out = {}
    out["dc"] = [r, be.solve(snap, abi.MARGIN_OLD), be.solve(snap, abi.MARGIN_SECOND_NEW)]
    h = abi.WindowHolder(snap)
    th = []
    for _ in range(30):
        t0 = time.perf_counter(); be.solve_raw(h, abi.MARGIN_OLD); th.append(time.perf_counter() - t0)
    be.close()
    if ql:
        bq = gf.Backend(0, options=o, so=ql)
        out["ql"] = [bq.solve(scn.window(0), abi.MARGIN_OLD), bq.solve(snap, abi.MARGIN_OLD), bq.solve(snap, abi.MARGIN_SECOND_NEW)]
        hq = abi.WindowHolder(snap)
        tq = []
        for _ in range(30):
            t0 = time.perf_counter(); bq.solve_raw(hq, abi.MARGIN_OLD); tq.append(time.perf_counter() - t0)
        bq.close()
    out["or"] = [orc.solve(scn.window(0), abi.MARGIN_OLD), orc.solve(snap, abi.MARGIN_OLD), orc.solve(snap, abi.MARGIN_SECOND_NEW)]
    line = "seed %d L %d: n = %s, host to host %.3f ms%s" % (seed, L, [x["prior"]["n"] for x in out["dc"]], np.median(th[5:]) * 1e3, (" (QL %.3f ms)" % (np.median(tq[5:]) * 1e3)) if ql else "")
    for k, (a) in enumerate(out["dc"]):
        J, r0 = a["prior"]["J0"], a["prior"]["r0"]
        A, b = J.T @ J, J.T @ r0
        G = J @ J.T
        orth = np.abs(G - np.diag(np.diag(G))).max() / max(np.abs(G).max(), 1e-300)
        worst["orth"] = max(worst["orth"], orth)
        for ref in ("ql", "or"):
            if ref not in out:
                continue
            Jr, rr = out[ref][k]["prior"]["J0"], out[ref][k]["prior"]["r0"]
            Ar, br = Jr.T @ Jr, Jr.T @ rr
            dA, db = np.abs(A - Ar).max() / np.abs(Ar).max(), np.abs(b - br).max() / max(np.abs(br).max(), 1.0)
            worst["A_" + ref] = max(worst["A_" + ref], dA); worst["b_" + ref] = max(worst["b_" + ref], db)
            line += "  [%d %s dA %.1e db %.1e r0^2 %.6e/%.6e]" % (k, ref, dA, db, r0 @ r0, rr @ rr)
    print(line, flush=True)
print("worst:", worst)
