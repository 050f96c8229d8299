import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Soak: N random windows (landmark count, wheel on/off, prior on/off, LiDAR block on/off, RGB-D constant landmarks, both
marginalisation flavours, and the four rare branches of tests/test_gpu_branches.py: every optional block free with subset
masks, frame 0 within a degree of gimbal lock, a forced mu retry, no IMU with Pose[0] constant) through the HIP library and
the CPU oracle; prints the largest deviations and every window whose
discrete outcome (iterations, accept / reject sequence, termination) differs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
N = int(os.environ.get("N", "40"))
_opt = abi.default_options()
_opt.marg_sqrt = int(os.environ.get("MARG_SQRT", "1"))      # (0: the device takes the reference's eigen square root of the new prior — divide & conquer, round 6)
be = gf.Backend(0, options=_opt)
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
    # the rare branches
    free_all = wheel and rng.random() < 0.2
    if free_all:
        snap.update(ex_cam_const=0, ex_wheel_const=0, ix_wheel_const=0, td_const=0, td_wheel_const=0,
                    ex_cam_mask=np.array([0, 0, 1, 0, 0, 0], np.uint8), ex_wheel_mask=np.array([0, 0, 1, 1, 1, 0], np.uint8),
                    ix_wheel=np.array([1.01, 0.99, 1.02]), td=0.002, td_wheel=-0.003)
    gimbal = (not with_prior) and (not partial) and rng.random() < 0.1
    if gimbal:
        q = synth.rot2q(synth.rz(rng.uniform(-3, 3)) @ synth.ry(np.deg2rad(rng.choice([-1, 1]) * rng.uniform(89.2, 89.9))) @ synth.rx(0.02))
        snap["pose"] = snap["pose"].copy()
        snap["pose"][0, 3:] = q / np.linalg.norm(q)
    no_imu = wheel and (not with_prior) and (not partial) and (not gimbal) and rng.random() < 0.1
    if no_imu:
        snap["imu"], snap["imu_frame"] = np.zeros((0, abi.IMU_DOUBLES)), np.zeros(0, np.int32)
        pc = np.zeros(abi.NFRAMES, np.uint8)
        pc[0] = 1
        snap["pose_const"] = pc
    # GNSS inside the window (round 3): observations of 3..12 satellites per frame from the scenario's true trajectory; one in five of
    # those windows too slow for the GNSS residual blocks (the lowspeed gate), whose frame-0 factors are still marginalised
    gnss = wheel and (not partial) and (not gimbal) and rng.random() < 0.25
    slow = gnss and rng.random() < 0.2
    if gnss:
        import gnss_window_cases as gw
        tru = gw.GnssTruth(scn, 1000 + i, n_per_frame=int(rng.integers(3, 13)), lat=float(rng.uniform(-60, 60)), lon=float(rng.uniform(-180, 180)))
        snap["gnss"], snap["gnss_state"] = tru.block(k0), tru.state(k0, 1000 + i)
        if slow:
            snap["speed_bias"] = np.array(snap["speed_bias"], float).copy()
            snap["speed_bias"][:, :2] *= 0.2
    retry = int(rng.integers(1, 5)) if rng.random() < 0.15 else 0
    if retry:      # gfbe_options.test_fail_chol_iter (test hook) on both sides
        oo = abi.default_options()
        oo.test_fail_chol_iter = retry
        ber = gf.Backend(device=0, options=oo)
        want, got = orc.with_options(test_fail_chol_iter=retry).solve(snap, flag), ber.solve(snap, flag)
        ber.close()
    else:
        want, got = orc.solve(snap, flag), be.solve(snap, flag)
    sw, sg = want["summary"], got["summary"]
    tag = "L=%d wheel=%d prior=%d lidar=%d rgbd=%d flag=%d partial=%d free=%d gimbal=%d noimu=%d retry=%d gnss=%d slow=%d" % (
        L, wheel, with_prior, lidar, rgbd, flag, partial, free_all, gimbal, no_imu, retry, gnss, slow)
    counts = globals().setdefault("counts", dict(free=0, gimbal=0, noimu=0, retry=0, gnss=0, gnss_slow=0))
    counts["free"] += free_all; counts["gimbal"] += gimbal; counts["noimu"] += no_imu; counts["retry"] += retry > 0
    counts["gnss"] += gnss; counts["gnss_slow"] += slow
    differs = (sw["iterations"], sw["accepted"], sw["termination"]) != (sg["iterations"], sg["accepted"], sg["termination"])
    if differs and gimbal:
        # (a gimbal-lock window is compared like the test compares it — under the two-iteration cap, below; over eight iterations of its
        #  violent re-convergence the last bits decide late accept / reject steps: counted, not a parity failure)
        globals().setdefault("gimbal_full_differs", []).append((i, sw["accepted"], sg["accepted"]))
    elif differs:
        bad.append((i, tag, sw["iterations"], sg["iterations"], sw["accepted"], sg["accepted"]))
        continue
    if gimbal:
        # frame 0 thrown 89 degrees off makes eight iterations of violent re-convergence in which last-bit differences grow
        # (tests/test_gpu_branches.py caps this case at two iterations): the discrete outcome and the first two costs are compared
        g = abs(np.array(sg["cost_history"][:3]) / np.array(sw["cost_history"][:3]) - 1).max() if not differs else 0.0
        wg = globals().setdefault("worst_gimbal", [0.0, 0.0, 0.0, 0.0])
        wg[0] = max(wg[0], g)
        if not differs:
            wg[1] = max(wg[1], abs(sg["final_cost"] / sw["final_cost"] - 1))
        # ... and the whole result under the two-iteration cap of the test: final cost and poses of both sides
        o2 = abi.default_options()
        o2.max_num_iterations = 2
        be2 = gf.Backend(0, options=o2)
        w2, g2 = orc.with_options(max_num_iterations=2).solve(snap, flag), be2.solve(snap, flag)
        be2.close()
        if (w2["summary"]["iterations"], w2["summary"]["accepted"]) != (g2["summary"]["iterations"], g2["summary"]["accepted"]):
            bad.append((i, tag + " (2-iteration cap)", w2["summary"]["iterations"], g2["summary"]["iterations"], w2["summary"]["accepted"], g2["summary"]["accepted"]))
        wg[2] = max(wg[2], abs(g2["summary"]["final_cost"] / w2["summary"]["final_cost"] - 1))
        wg[3] = max(wg[3], float(np.abs(g2["state"]["pose"][:, :3] - w2["state"]["pose"][:, :3]).max()))
        continue
    dev = abs(sg["final_cost"] / sw["final_cost"] - 1)
    if dev > 1e-6:
        print("  window %d: final cost %.3e rel off (%s; iterations %d, accepted %s, termination %d)" % (i, dev, tag, sw["iterations"], sw["accepted"], sw["termination"]))
    rare = free_all or gimbal or no_imu or retry
    key = "gnss" if gnss else ("rare" if rare else "std")
    wr = globals().setdefault("worst_by", dict(rare=0.0, std=0.0, gnss=0.0))
    wr[key] = max(wr[key], dev)
    worst["cost"] = max(worst["cost"], dev)
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
print("rare branches drawn:", counts, "; largest final-cost deviation: standard windows %.2e, rare-branch windows %.2e, GNSS windows %.2e" % (worst_by["std"], worst_by["rare"], worst_by["gnss"]))
for b in bad:
    print("  DIFFERS:", b)
if "gimbal_full_differs" in globals():
    print("gimbal-lock windows whose accept / reject sequence over the full eight iterations differs (their two-iteration runs are compared above): %s" % gimbal_full_differs)
if "worst_gimbal" in globals():
    print("gimbal-lock windows: costs of the first two iterations %.2e rel, final cost after eight %.2e rel (not settled, see the source); "
          "under the test's two-iteration cap: final cost %.2e rel, poses %.2e m" % tuple(worst_gimbal))
print("largest deviations: final cost %.2e rel, ATE %.2e m, rotation %.2e rad, inverse depth %.2e rel, prior J0^T J0 %.2e rel"
      % (worst["cost"], worst["ate"], worst["rot"], worst["lam"], worst["prior"]))
