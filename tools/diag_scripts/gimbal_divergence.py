import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests"), _os.path.join(_r, "tools")]
"""Why ONE gimbal-lock window of the soak (tools/diag_soak.py, window 70 of the N = 400 / 800 runs) takes another accept / reject step
at iteration 7 on the device than in the oracle (VERDICT round 5, Weak 3: "unexplained beyond chaotic re-convergence").
The window is drawn again exactly as the soak draws it (the same generator, the same draws); then
  (a) the ORACLE is run against ITSELF on inputs that differ in the last bit of one number (five such perturbations), and
  (b) the device (when a GPU is present) against the oracle,
and the cost after every iteration is printed for all of them relative to the unperturbed oracle. If (a) departs from the oracle as far
and as early as (b) does, the difference between the two implementations is the window's sensitivity to its last bits and not a
difference of algorithm.   WIN=70 python tools/diag_scripts/gimbal_divergence.py"""
import os, sys
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
WIN = int(os.environ.get("WIN", "70"))
orc = oracle_lib.load()
rng = np.random.default_rng(2026)
snap = flag = None
for i in range(WIN + 1):      # tools/diag_soak.py's draws, in its order (the solves that draw nothing are skipped)
    L = int(rng.choice([60, 200, 700, 2000, 3500]))
    wheel, with_prior, lidar, rgbd = bool(rng.integers(2)), bool(rng.integers(2)), rng.random() < 0.3, rng.random() < 0.3
    flag = int(rng.choice([abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW, abi.MARGIN_NONE]))
    scn = synth.Scenario(seed=1000 + i, n_landmarks=L, use_wheel=wheel)
    snap = scn.window(0)
    k0 = 0
    if with_prior:
        if rgbd or i == WIN:      # (the number of landmarks of the second window decides how many numbers the RGB-D mask draws)
            r0 = orc.solve(snap, abi.MARGIN_OLD)
            snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
        k0 = 1
    if rgbd:
        fc = np.zeros(len(snap["para_feature"]), np.uint8)
        fc[rng.random(len(fc)) < 0.5] = 1
        snap["feature_const"] = fc
    if lidar:
        nl = int(rng.choice([50, 800, 2000]))
        if i == WIN:
            snap["lio"] = synth.lidar_block(scn, k0, n=nl, seed=i, outliers=0.05)
    partial = (not with_prior) and rng.random() < 0.25
    if partial:
        fcn = int(rng.integers(2, abi.WINDOW_SIZE))
        keep = snap["vis_imu_j"] <= fcn
        for k in list(snap):
            if k.startswith("vis_"):
                snap[k] = snap[k][keep]
        snap["frame_count"] = fcn
        snap["imu"], snap["imu_frame"] = snap["imu"][:fcn], snap["imu_frame"][:fcn]
        if wheel:
            snap["wheel"], snap["wheel_frame"] = snap["wheel"][:fcn], snap["wheel_frame"][:fcn]
        if lidar and "lio" in snap:
            snap["lio"]["frame"] = fcn
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
    gnss = wheel and (not partial) and (not gimbal) and rng.random() < 0.25
    slow = gnss and rng.random() < 0.2
    if gnss:
        n_per_frame, lat, lon = int(rng.integers(3, 13)), float(rng.uniform(-60, 60)), float(rng.uniform(-180, 180))
    retry = int(rng.integers(1, 5)) if rng.random() < 0.15 else 0
print("window %d: L=%d wheel=%d prior=%d lidar=%d rgbd=%d flag=%d partial=%d free=%d gimbal=%d noimu=%d retry=%d gnss=%d"
      % (WIN, L, wheel, with_prior, lidar, rgbd, flag, partial, free_all, gimbal, no_imu, retry, gnss))
assert not retry and not gnss, "this script replays the plain / gimbal-lock windows of the soak"

want = orc.solve(snap, flag)
sw = want["summary"]
h0 = np.array(sw["cost_history"][: sw["iterations"] + 1])
print("oracle            accepted %s" % sw["accepted"][: sw["iterations"] + 1])
rows = []


def rel_hist(s):
    h = np.array(s["cost_history"][: s["iterations"] + 1])
    m = min(len(h), len(h0))
    return np.abs(h[:m] / h0[:m] - 1)


# (a) the oracle on inputs one unit in the last place away: a position coordinate, a quaternion entry (renormalised by the solver's
#     first retraction anyway), a velocity, an inverse depth, an observation
perts = [("pose[5].x + 1 ulp", "pose", (5, 0)), ("pose[3].qz + 1 ulp", "pose", (3, 5)), ("speed_bias[7].vx + 1 ulp", "speed_bias", (7, 0)),
         ("para_feature[0] + 1 ulp", "para_feature", (0,)), ("pose[9].z + 1 ulp", "pose", (9, 2))]
for name, key, idx in perts:
    s2 = dict(snap)
    a = np.array(s2[key], float).copy()
    a[idx] = np.nextafter(a[idx], np.inf)
    s2[key] = a
    r = orc.solve(s2, flag)["summary"]
    rows.append(("oracle, " + name, r["accepted"][: r["iterations"] + 1], rel_hist(r)))
# (b) the device
try:
    import torch
    if torch.cuda.is_available():
        be = gf.Backend(0)
        g = be.solve(snap, flag)["summary"]
        rows.append(("device", g["accepted"][: g["iterations"] + 1], rel_hist(g)))
        be.close()
except Exception as e:      # (no GPU here: the oracle's half still says how sensitive the window is)
    print("device not run:", str(e)[:80])
print("cost after iteration k relative to the unperturbed oracle's (0 = the same bits):")
print("%-36s %s" % ("", "  ".join("it %d    " % k for k in range(len(h0)))))
for name, acc, rh in rows:
    print("%-36s %s   accepted %s" % (name, "  ".join("%.1e" % v for v in rh), acc))
print("oracle's cost history:", "  ".join("%.6g" % v for v in h0))
