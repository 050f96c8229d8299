"""Soak of the batch paths: N random windows (as in diag_soak.py) solved (a) one by one, (b) in one batch of N (two halves,
second-stream dense factors, one-workgroup k_visblock) and (c) in batches of 7 — every output must be bit-identical."""
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
for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
    t0 = time.time()
    single = [be.solve(s, flag) for s in snaps]
    big = be.solve_batch(snaps, flag)
    small = []
    for k in range(0, N, 7):
        small += be.solve_batch(snaps[k:k + 7], flag)
    bad_big = [i for i in range(N) if not same(single[i], big[i])]
    bad_small = [i for i in range(N) if not same(single[i], small[i])]
    print("flag %d: %d windows, %.0f s: batch of %d differs from single solves in %d windows, batches of 7 in %d" % (flag, N, time.time() - t0, N, len(bad_big), len(bad_small)), bad_big[:5], bad_small[:5])

# ---- which windows differ, and by how much
flag = abi.MARGIN_OLD
single = [be.solve(s, flag) for s in snaps]
for B in (40,):
    got = []
    for k in range(0, N, B):
        got += be.solve_batch(snaps[k:k + B], flag) if len(snaps[k:k + B]) >= 32 else [None] * len(snaps[k:k + B])
    rows = []
    for i in range(N):
        if got[i] is None:
            continue
        d = not same(single[i], got[i])
        s = snaps[i]
        rows.append((d, len(s.get("wheel", [])) > 0, s.get("prior") is not None, "lio" in s, "feature_const" in s,
                     np.abs(single[i]["state"]["pose"] - got[i]["state"]["pose"]).max()))
    rows = np.array(rows, dtype=float)
    print("batches of %d: %d of %d differ; among differing: wheel %.2f prior %.2f lio %.2f const %.2f; among identical: wheel %.2f prior %.2f lio %.2f const %.2f; max |dpose| %.1e"
          % (B, int(rows[:, 0].sum()), len(rows), *rows[rows[:, 0] == 1][:, 1:5].mean(axis=0), *rows[rows[:, 0] == 0][:, 1:5].mean(axis=0), rows[:, 5].max()))
