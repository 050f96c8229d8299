import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""The chain kernels side by side on ONE box (round 5): per-launch time of the solve kernel for one window and for a 2048-window
part — gfbe_options.solve_kernel = 2 (one-ended chain, k_solve_chain) / 3 (two-ended, k_solve_chain_tw) / 1 (monolithic) —, the
single-window solve times (resident and host to host), and the throughput of a 4 x 2048 batch. GFBE_LIB selects a variant library."""
import os, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth

scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
be0 = gf.Backend(0)
r = be0.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
be0.close()
ref = None
for kernel in [int(k) for k in os.environ.get("KERNELS", "2,3,1").split(",")]:
    o = abi.default_options(); o.solve_kernel = kernel
    be = gf.Backend(0, options=o)
    one = be.batch_upload([snap])
    for _ in range(20): one.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter(); one.solve(abi.MARGIN_OLD); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res = one.download()[0]
    h = abi.WindowHolder(snap)
    th = []
    for _ in range(120):
        t0 = time.perf_counter(); be.solve_raw(h, abi.MARGIN_OLD); th.append(time.perf_counter() - t0)
    be.profile_enable(True); be.profile_reset()
    for _ in range(5): one.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    prof1 = {p["name"]: 1e3 * p["total_ms"] / max(p["launches"], 1) for p in be.profile() if p["launches"]}
    be.profile_enable(False)
    one.solve(abi.MARGIN_OLD); torch.cuda.synchronize()
    tm = one.debug_timing(0)
    ph = [("prologue", 0, 1), ("build", 1, 2), ("pipeline", 2, 15), ("dense chol", 15, 3), ("dense backsub", 3, 16), ("chain backsub", 16, 4), ("gram", 4, 5), ("total", 0, 5)]
    if kernel == 1:      # (the monolithic kernel stamps no pipeline: its factorisation runs from stamp 2 to 3, its back-substitution to 4)
        ph = [("prologue", 0, 1), ("build", 1, 2), ("cholesky", 2, 3), ("backsub", 3, 4), ("gram", 4, 5), ("total", 0, 5)]

    def span(a, b):      # a phase whose stamps this kernel did not write (a slot left at zero, or one of an earlier run) is not a time
        dt = (tm[b] - tm[a]) * 0.01
        return "%.2f" % dt if (tm[a] > 0 and tm[b] >= tm[a] and dt < 1e5) else "n/a"
    print("   phases (us, last iteration): " + "  ".join("%s %s" % (n, span(a, b)) for n, a, b in ph), flush=True)
    one.free()
    if ref is None: ref = res
    print("solve_kernel %d: one window resident %.4f ms (p10 %.4f) host-to-host %.4f ms | k_solve %.1f us per launch (iter0 %.1f) | final cost %.12g, dpos vs first kernel %.2e" %
          (kernel, np.median(ts) * 1e3, np.percentile(ts, 10) * 1e3, np.median(th[20:]) * 1e3, prof1.get("k_solve", 0), prof1.get("k_solve_iter0", 0),
           res["summary"]["final_cost"], np.abs(res["state"]["pose"] - ref["state"]["pose"]).max()), flush=True)
    if kernel != 1 and int(os.environ.get("BIG", "1")):
        B = int(os.environ.get("B", "2048"))
        b = be.batch_upload([snap] * B)
        b.solve(abi.MARGIN_OLD); torch.cuda.synchronize()
        be.profile_enable(True); be.profile_reset()
        for _ in range(2): b.solve(abi.MARGIN_OLD)
        torch.cuda.synchronize()
        prof = {p["name"]: 1e3 * p["total_ms"] / max(p["launches"], 1) for p in be.profile() if p["launches"]}
        be.profile_enable(False)
        t0 = time.perf_counter()
        for _ in range(3): b.solve(abi.MARGIN_OLD)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("   B=%d: k_solve %.1f us per launch (iter0 %.1f); whole batch %.2f ms = %.0f solves/s" % (B, prof.get("k_solve", 0), prof.get("k_solve_iter0", 0), dt * 1e3, B / dt), flush=True)
        b.free()
    be.close()
