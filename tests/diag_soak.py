"""Soak: N random windows (landmark count, wheel on/off, prior on/off, LiDAR block on/off, RGB-D constant landmarks, both
marginalisation flavours) through the HIP library and the CPU oracle; prints the largest deviations and every window whose
discrete outcome (iterations, accept / reject sequence, termination) differs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
N = int(os.environ.get("N", "40"))
be = gf.Backend(0)
orc = oracle_lib.load()
rng = np.random.default_rng(2026)
worst = dict(cost=0.0, ate=0.0, rot=0.0, lam=0.0, prior=0.0)
bad = []
t0 = time.time()
for i in range(N):
    L = int(rng.choice([60, 200, 700, 2000, 3500]))
    wheel, with_prior, lidar, rgbd = bool(rng.integers(2)), bool(rng.integers(2)), rng.random() < 0.3, rng.random() < 0.3
    flag = int(rng.choice([abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW, abi.MARGIN_NONE]))
    scn = synth.Scenario(seed=1000 + i, n_landmarks=L, use_wheel=wheel)
    snap = scn.window(0)
    k0 = 0
    if with_prior:
        r0 = orc.solve(snap, abi.MARGIN_OLD)
        snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
        k0 = 1
    if rgbd:
        fc = np.zeros(len(snap["para_feature"]), np.uint8)
        fc[rng.random(len(fc)) < 0.5] = 1
        snap["feature_const"] = fc
    if lidar:
        snap["lio"] = synth.lidar_block(scn, k0, n=int(rng.choice([50, 800, 2000])), seed=i, outliers=0.05)
    partial = (not with_prior) and rng.random() < 0.25      # a window that is still filling up (frame_count < WINDOW_SIZE): no marginalisation
    if partial:
        fc = int(rng.integers(2, abi.WINDOW_SIZE))
        keep = snap["vis_imu_j"] <= fc
        for k in list(snap):
            if k.startswith("vis_"):
                snap[k] = snap[k][keep]
        snap["frame_count"] = fc
        snap["imu"], snap["imu_frame"] = snap["imu"][:fc], snap["imu_frame"][:fc]
        if wheel:
            snap["wheel"], snap["wheel_frame"] = snap["wheel"][:fc], snap["wheel_frame"][:fc]
        if lidar:
            snap["lio"]["frame"] = fc
    want, got = orc.solve(snap, flag), be.solve(snap, flag)
    sw, sg = want["summary"], got["summary"]
    tag = "L=%d wheel=%d prior=%d lidar=%d rgbd=%d flag=%d partial=%d" % (L, wheel, with_prior, lidar, rgbd, flag, partial)
    if (sw["iterations"], sw["accepted"], sw["termination"]) != (sg["iterations"], sg["accepted"], sg["termination"]):
        bad.append((i, tag, sw["iterations"], sg["iterations"], sw["accepted"], sg["accepted"]))
        continue
    worst["cost"] = max(worst["cost"], abs(sg["final_cost"] / sw["final_cost"] - 1))
    worst["ate"] = max(worst["ate"], np.sqrt(((got["state"]["pose"][:, :3] - want["state"]["pose"][:, :3]) ** 2).sum(axis=1).mean()))
    for f in range(abi.NFRAMES):
        dq = synth.qmul(synth.qinv(want["state"]["pose"][f, 3:]), got["state"]["pose"][f, 3:])
        worst["rot"] = max(worst["rot"], 2 * np.linalg.norm(dq[:3]))
    if len(want["feature"]):
        worst["lam"] = max(worst["lam"], np.abs(got["feature"] / want["feature"] - 1).max())
    if want["prior"] is not None:
        A, Ag = want["prior"]["J0"].T @ want["prior"]["J0"], got["prior"]["J0"].T @ got["prior"]["J0"]
        worst["prior"] = max(worst["prior"], np.abs(A - Ag).max() / np.abs(A).max())
print("%d random windows in %.0f s; discrete outcome differs in %d" % (N, time.time() - t0, len(bad)))
for b in bad:
    print("  DIFFERS:", b)
print("largest deviations: final cost %.2e rel, ATE %.2e m, rotation %.2e rad, inverse depth %.2e rel, prior J0^T J0 %.2e rel"
      % (worst["cost"], worst["ate"], worst["rot"], worst["lam"], worst["prior"]))
