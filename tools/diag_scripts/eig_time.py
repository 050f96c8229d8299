import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""marg_sqrt = 0 (the reference's eigen square root of the new prior) for one 2k-landmark window: the call's time resident / host to host,
the marginalisation kernels' times, and the new prior against marg_sqrt = 1's (information J0^T J0, J0^T r0)."""
import time
import numpy as np
import torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
be0 = gf.Backend(0)
r = be0.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
be0.close()
pri = {}
for mode in (1, 0):
    o = abi.default_options(); o.marg_sqrt = mode
    be = gf.Backend(0, options=o)
    one = be.batch_upload([snap])
    for _ in range(5): one.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); one.solve(abi.MARGIN_OLD); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res = one.download()[0]
    pri[mode] = res["prior"]
    h = abi.WindowHolder(snap)
    th = []
    for _ in range(40):
        t0 = time.perf_counter(); be.solve_raw(h, abi.MARGIN_OLD); th.append(time.perf_counter() - t0)
    be.profile_enable(True); be.profile_reset()
    for _ in range(5): one.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    pm = {p["name"]: 1e3 * p["total_ms"] / max(p["launches"], 1) for p in be.profile() if p["launches"]}
    print("marg_sqrt %d: resident %.3f ms, host to host %.3f ms; marginalize %.1f us per solve; prior n %d" % (mode, np.median(ts) * 1e3, np.median(th[5:]) * 1e3, pm.get("marginalize", 0), res["prior"]["n"]))
    one.free(); be.close()
A0, A1 = pri[0]["J0"].T @ pri[0]["J0"], pri[1]["J0"].T @ pri[1]["J0"]
b0, b1 = pri[0]["J0"].T @ pri[0]["r0"], pri[1]["J0"].T @ pri[1]["r0"]
print("eigen vs LDL^T prior: max|dA| / max|A| %.2e   max|db| / max|b| %.2e   rows of J0 with weight: %d / %d" % (
    np.abs(A0 - A1).max() / np.abs(A1).max(), np.abs(b0 - b1).max() / max(np.abs(b1).max(), 1.0),
    int((np.abs(pri[0]["J0"]).sum(axis=1) > 0).sum()), int((np.abs(pri[1]["J0"]).sum(axis=1) > 0).sum())))
# phase stamps of the eigen-decomposition (diagnostics build only: GFBE_LIB=.../libgfbe_diag.so)
o = abi.default_options(); o.marg_sqrt = 0
be = gf.Backend(0, options=o)
one = be.batch_upload([snap])
for _ in range(3): one.solve(abi.MARGIN_OLD)
torch.cuda.synchronize()
tm = one.debug_timing(1)
if tm[11] > tm[8] > 0 and tm[12] == 0:
    print("tridiag_ql_eig: tridiagonalise %.1f us, accumulate Q %.1f us, divide & conquer + Q_dc^T Z %.1f us" % ((tm[9] - tm[8]) * 0.01, (tm[10] - tm[9]) * 0.01, (tm[11] - tm[10]) * 0.01))
    print("  tridiag_dc, summed over its levels (us): leaves %.1f | z + rank sort %.1f | deflation scan %.1f | rotations + secular roots %.1f | zhat %.1f | norms %.1f | vectors %.1f | Q <- Q V %.1f" % tuple(tm[13 + q] * 0.01 for q in range(8)))
elif tm[11] > tm[8] > 0:
    print("tridiag_ql_eig: tridiagonalise %.1f us, accumulate Q %.1f us, QL %.1f us (%d sweeps: %.2f us per sweep)" % (
        (tm[9] - tm[8]) * 0.01, (tm[10] - tm[9]) * 0.01, (tm[11] - tm[10]) * 0.01, int(tm[12]), (tm[11] - tm[10]) * 0.01 / max(tm[12], 1)))
