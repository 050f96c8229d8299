import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Soak of the batch paths: N random windows (as in diag_soak.py) solved (a) one by one, (b) in batches of 7, (c) in one batch of N
and (d) in batches of 40: (a) == (b) and (c) == (d) bit for bit; (a) vs (c) — the small-batch and the throughput kernel set — at
tolerance with identical discrete outcomes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
N = int(os.environ.get("N", "150"))
be = gf.Backend(0)
rng = np.random.default_rng(77)
snaps = []
for i in range(N):
    L = int(rng.choice([60, 200, 700, 2000]))
    scn = synth.Scenario(seed=5000 + i, n_landmarks=L, use_wheel=bool(rng.integers(2)))
    snap = scn.window(0)
    if rng.random() < 0.5:
        r0 = be.solve(snap, abi.MARGIN_OLD)
        snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    if rng.random() < 0.3:
        fc = np.zeros(len(snap["para_feature"]), np.uint8); fc[rng.random(len(fc)) < 0.5] = 1
        snap["feature_const"] = fc
    if rng.random() < 0.3:
        snap["lio"] = synth.lidar_block(scn, 1 if snap.get("prior") is not None else 0, n=int(rng.choice([50, 800, 2000])), seed=i, outliers=0.05)
    snaps.append(snap)
def same(a, b):
    if a["summary"] != b["summary"] or not np.array_equal(a["feature"], b["feature"]):
        return False
    fa, fb = abi.flat_state(a["state"]), abi.flat_state(b["state"])
    for k in fa:
        if not np.array_equal(np.asarray(fa[k]), np.asarray(fb[k])):
            return False
    if (a["prior"] is None) != (b["prior"] is None):
        return False
    return a["prior"] is None or all(np.array_equal(a["prior"][k], b["prior"][k]) for k in ("J0", "r0", "x0", "block_id"))
def deviation(a, b):
    sa, sb = a["summary"], b["summary"]
    if (sa["iterations"], sa["accepted"], sa["termination"]) != (sb["iterations"], sb["accepted"], sb["termination"]):
        return None
    return max(abs(sb["final_cost"] / sa["final_cost"] - 1), np.abs(a["state"]["pose"] - b["state"]["pose"]).max())
# The guarantees since round 3: a window is bit-reproducible and independent of its neighbours inside each
# kernel set — small batches (< 32 windows: single solves == batches of 7) and throughput batches (one batch of N == batches of 40) —
# and the two sets agree at tolerance (k_schur sums a window's landmark tiles in 4 start-frame groups instead of 22).
for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
    t0 = time.time()
    single = [be.solve(s, flag) for s in snaps]
    big = be.solve_batch(snaps, flag)
    small, mid = [], []
    for k in range(0, N, 7):
        small += be.solve_batch(snaps[k:k + 7], flag)
    for k in range(0, N, 40):
        chunk = snaps[k:k + 40]
        mid += be.solve_batch(chunk, flag) if len(chunk) >= 32 else [None] * len(chunk)
    bad_small = [i for i in range(N) if not same(single[i], small[i])]
    bad_mid = [i for i in range(N) if mid[i] is not None and not same(big[i], mid[i])]
    devs = [deviation(single[i], big[i]) for i in range(N)]
    print("flag %d: %d windows, %.0f s: batches of 7 differ from single solves in %d windows; batches of 40 differ from the batch of %d in %d windows; "
          "single vs throughput kernel set: discrete outcome differs in %d, largest deviation (final cost rel / pose abs) %.2e"
          % (flag, N, time.time() - t0, len(bad_small), N, len(bad_mid), sum(d is None for d in devs), max(d for d in devs if d is not None)), bad_small[:5], bad_mid[:5])
