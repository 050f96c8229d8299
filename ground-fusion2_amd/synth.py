"""Synthetic sliding-window generator: seeded ground-robot trajectory, IMU / wheel
streams, landmark tracks -> the *window snapshot* one Estimator::optimization() call reads.

Pure numpy, no oracle, no GPU: this is input generation for tests and bench.py. The mid-point
pre-integration below is an independent numpy statement of
  IntegrationBase::midPointIntegration       vins_estimator/src/factor/integration_base.h:63-137
  WheelIntegrationBase::midPointIntegration  vins_estimator/src/factor/wheel_integration_base.h:67-146
used to cross-check both the CPU oracle and the HIP kernels (tests/).
Constants come from Ground-Fusion++/config/realsense/m3dgr.yaml (lines cited inline).
"""
import numpy as np

from . import abi

G_NORM = 9.7944                     # m3dgr.yaml:117
ACC_N, GYR_N = 1.2374091609523514e-02, 3.0032654435730201e-03   # :113-114
ACC_W, GYR_W = 1.9218003442176448e-04, 5.4692100664858005e-05   # :115-116
VEL_N_WHEEL, GYR_N_WHEEL = 0.01, 0.004                          # :121-123
FOCAL = 600.0                       # parameters.h:23
BODY_T_CAM0 = np.array([[0.99957087, 0.00215313, 0.02921355, 0.03668114],       # m3dgr.yaml:44-51
                        [-0.00192891, 0.99996848, -0.00770122, -0.00477653],
                        [-0.02922921, 0.00764156, 0.99954353, 0.0316039],
                        [0, 0, 0, 1.0]])
BODY_T_WHEEL = np.array([[4.2873564019253907e-02, -9.9906999607154057e-01, 4.5826256555663858e-03, 1.0000278019634017e-00],
                         [2.3548883729155812e-02, -3.5750257528033291e-03, -9.9971629438855181e-01, 0.0477569625897234e-01],
                         [9.9880293731215963e-01, 4.2969316267296165e-02, 2.3373709079293481e-02, 2.0902387796334685e-01],
                         [0, 0, 0, 1.0]])                                        # m3dgr.yaml:76-85
KF_DT = 0.1                         # freq: 10 (m3dgr.yaml:102)
IMU_HZ, WHEEL_HZ = 200, 50


# ------------------------------------------------------------------ small SO(3) toolbox (x y z w)
def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):
    return np.array([-q[0], -q[1], -q[2], q[3]]) / np.dot(q, q)


def qrot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rot2q(R):
    """Rotation matrix -> unit quaternion (x y z w), w >= 0 branch first (trace method)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[i] = 0.5 * s
    s = 0.5 / s
    q[3] = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q


def so3_exp(w):
    th = np.linalg.norm(w)
    if th < 1e-10:
        return np.array([0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0])
    return np.concatenate([np.sin(0.5 * th) / th * w, [np.cos(0.5 * th)]])


def so3_log(q):
    n = np.linalg.norm(q[:3])
    if n < 1e-10:
        return 2.0 / q[3] * q[:3]
    return 2.0 * np.arctan(n / q[3]) / n * q[:3]


def right_jac(phi):
    n2 = float(phi @ phi)
    h = skew(phi)
    if n2 > 1e-10:
        n = np.sqrt(n2)
        return np.eye(3) - h * (1 - np.cos(n)) / n2 + h @ h * (n - np.sin(n)) / (n2 * n)
    return np.eye(3) - h / 2 + h @ h / 6


def rz(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


def ry(a):
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1.0, 0], [-np.sin(a), 0, np.cos(a)]])


def rx(a):
    return np.array([[1.0, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])


# body frame = camera convention (x right, y down, z forward): see body_T_wheel above, whose
# wheel-x (forward) column is ~ body z.
R0_BODY = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])


# ------------------------------------------------------------------ numpy mid-point pre-integration
def preintegrate_imu_np(samples, first, lin_ba, lin_bg, noise=(ACC_N, GYR_N, ACC_W, GYR_W)):
    """samples [n,7] = dt, acc(3), gyr(3); first = acc_0, gyr_0 (6). Returns the 467-double record."""
    acc_0, gyr_0 = first[:3].copy(), first[3:].copy()
    dp, dv, dq = np.zeros(3), np.zeros(3), np.array([0, 0, 0, 1.0])
    jac, cov = np.eye(15), np.zeros((15, 15))
    an, gn, aw, gw = noise
    N = np.diag(np.repeat([an * an, gn * gn, an * an, gn * gn, aw * aw, gw * gw], 3))
    sum_dt = 0.0
    I3 = np.eye(3)
    for row in samples:
        dt, acc_1, gyr_1 = row[0], row[1:4], row[4:7]
        un_acc_0 = qrot(dq) @ (acc_0 - lin_ba)
        un_gyr = 0.5 * (gyr_0 + gyr_1) - lin_bg
        rq = qmul(dq, np.array([un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0]))
        un_acc_1 = qrot(rq) @ (acc_1 - lin_ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        Rw, Ra0, Ra1 = skew(un_gyr), skew(acc_0 - lin_ba), skew(acc_1 - lin_ba)
        Rd, Rr = qrot(dq), qrot(rq)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * Rd @ Ra0 * dt * dt - 0.25 * Rr @ Ra1 @ (I3 - Rw * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (Rd + Rr) * dt * dt
        F[0:3, 12:15] = -0.25 * Rr @ Ra1 * dt * dt * -dt
        F[3:6, 3:6] = I3 - Rw * dt
        F[3:6, 12:15] = -I3 * dt
        F[6:9, 3:6] = -0.5 * Rd @ Ra0 * dt - 0.5 * Rr @ Ra1 @ (I3 - Rw * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (Rd + Rr) * dt
        F[6:9, 12:15] = -0.5 * Rr @ Ra1 * dt * -dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * Rd * dt * dt
        V[0:3, 3:6] = 0.25 * -Rr @ Ra1 * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * Rd * dt
        V[6:9, 3:6] = 0.5 * -Rr @ Ra1 * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        jac = F @ jac
        cov = F @ cov @ F.T + V @ N @ V.T
        dp, dv = rp, rv
        dq = rq / np.linalg.norm(rq)
        sum_dt += dt
        acc_0, gyr_0 = acc_1, gyr_1
    return np.concatenate([[sum_dt], dp, dq, dv, lin_ba, lin_bg, jac.ravel(), cov.ravel()])


def preintegrate_wheel_np(samples, first, lin, noise=(VEL_N_WHEEL, GYR_N_WHEEL)):
    """samples [n,7] = dt, vel(3), gyr(3); first = vel_0, gyr_0; lin = sx, sy, sw, td. 78 doubles."""
    vel_0, gyr_0 = first[:3].copy(), first[3:].copy()
    sx, sy, sw, td = lin
    sv = np.diag([sx, sy, 1.0])
    dp, dq = np.zeros(3), np.array([0, 0, 0, 1.0])
    jac, cov = np.zeros((6, 3)), np.zeros((6, 6))
    vn, gn = noise
    N = np.diag(np.repeat([vn * vn, gn * gn, vn * vn, gn * gn], 3))
    sum_dt = 0.0
    vel_1, gyr_1 = vel_0, gyr_0
    I1, I2 = np.diag([1.0, 0, 0]), np.diag([0, 1.0, 0])
    for row in samples:
        dt, vel_1, gyr_1 = row[0], row[1:4], row[4:7]
        un_vel_0 = qrot(dq) @ sv @ vel_0
        un_gyr = 0.5 * sw * (gyr_0 + gyr_1)
        ddq = np.array([un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0])
        rq = qmul(dq, ddq)
        un_vel_1 = qrot(rq) @ sv @ vel_1
        rp = dp + 0.5 * (un_vel_0 + un_vel_1) * dt
        Rv0, Rv1 = skew(sv @ vel_0), skew(sv @ vel_1)
        Rd, Rr, Rdd = qrot(dq), qrot(rq), qrot(ddq)
        F = np.zeros((6, 6))
        F[0:3, 0:3] = np.eye(3)
        F[0:3, 3:6] = -0.5 * dt * (Rd @ Rv0 + Rr @ Rv1 @ Rdd.T)
        F[3:6, 3:6] = Rdd.T
        Jr = right_jac(un_gyr * dt)
        V = np.zeros((6, 12))
        V[0:3, 0:3] = 0.5 * dt * Rd @ sv
        V[0:3, 3:6] = -0.25 * dt * dt * Rr @ Rv1 @ Jr
        V[0:3, 6:9] = 0.5 * dt * Rr @ sv
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * Jr * sw * dt
        V[3:6, 9:12] = 0.5 * Jr * sw * dt
        jac[0:3, 0] += 0.5 * (Rd @ I1 @ vel_0 + Rr @ I1 @ vel_1) * dt
        jac[0:3, 1] += 0.5 * (Rd @ I2 @ vel_0 + Rr @ I2 @ vel_1) * dt
        last = jac[3:6, 2].copy()
        jac[3:6, 2] += Jr @ (0.5 * (gyr_0 + gyr_1) * dt)
        jac[0:3, 2] += 0.5 * (Rd @ skew(last) @ sv @ vel_0 + Rr @ skew(jac[3:6, 2]) @ sv @ vel_1) * dt
        cov = F @ cov @ F.T + V @ N @ V.T
        dp = rp
        dq = rq / np.linalg.norm(rq)
        sum_dt += dt
        vel_0, gyr_0 = vel_1, gyr_1
    return np.concatenate([[sum_dt], dp, dq, [sx, sy, sw, td], first[:3], first[3:], vel_1, gyr_1,
                           jac.ravel(), cov.ravel()])


# ------------------------------------------------------------------ scenario
class Scenario:
    """A 12-keyframe ground-robot run; window k covers keyframes k..k+10 (k = 0, 1)."""

    def __init__(self, seed=20250708, n_landmarks=2000, use_wheel=True, noise=True, n_kf=12,
                 estimate_wheel_extrinsic=True):
        self.seed = seed
        self.L = n_landmarks
        self.use_wheel = use_wheel
        self.noise = noise
        self.n_kf = n_kf
        self.est_wheel_ex = estimate_wheel_extrinsic
        rng = np.random.default_rng(seed)
        self.rng = rng
        sc = 1.0 if noise else 0.0
        # smooth roll / pitch / z wobble (amplitudes 0.5 deg, 0.5 deg, 5 mm)
        self.wob = dict(ar=np.deg2rad(0.5) * rng.normal() * sc, ap=np.deg2rad(0.5) * rng.normal() * sc,
                        az=0.005 * rng.normal() * sc, fr=rng.uniform(0.3, 1.0), fp=rng.uniform(0.3, 1.0),
                        fz=rng.uniform(0.3, 1.0), phr=rng.uniform(0, 6.28), php=rng.uniform(0, 6.28),
                        phz=rng.uniform(0, 6.28))
        self.ba = rng.normal(0, 0.02, 3)
        self.bg = rng.normal(0, 0.002, 3)
        self.ba_est = self.ba + sc * rng.normal(0, 0.005, 3)
        self.bg_est = self.bg + sc * rng.normal(0, 0.0005, 3)
        self.ric, self.tic = BODY_T_CAM0[:3, :3].copy(), BODY_T_CAM0[:3, 3].copy()
        self.rio, self.tio = BODY_T_WHEEL[:3, :3].copy(), BODY_T_WHEEL[:3, 3].copy()
        # orthonormalise the yaml rotations (they are rounded to ~1e-8)
        self.ric = qrot(rot2q(self.ric) / np.linalg.norm(rot2q(self.ric)))
        self.rio = qrot(rot2q(self.rio) / np.linalg.norm(rot2q(self.rio)))
        self.kf_t = np.arange(n_kf) * KF_DT
        self._make_inertial()

    # --- ground truth trajectory
    def pose(self, t):
        w = self.wob
        yaw = 0.2 * t
        p = np.array([5 * np.sin(yaw), 5 * (1 - np.cos(yaw)), w["az"] * np.sin(2 * np.pi * w["fz"] * t + w["phz"])])
        roll = w["ar"] * np.sin(2 * np.pi * w["fr"] * t + w["phr"])
        pitch = w["ap"] * np.sin(2 * np.pi * w["fp"] * t + w["php"])
        R = rz(yaw) @ ry(pitch) @ rx(roll) @ R0_BODY
        return p, R

    def kinematics(self, t, h=1e-4):
        p0, R0 = self.pose(t)
        pm, Rm = self.pose(t - h)
        pp, Rp = self.pose(t + h)
        v = (pp - pm) / (2 * h)
        a = (pp - 2 * p0 + pm) / (h * h)
        om = so3_log(rot2q(Rm.T @ Rp)) / (2 * h)      # body angular velocity
        return p0, R0, v, a, om

    def _make_inertial(self):
        rng, sc = self.rng, (1.0 if self.noise else 0.0)
        G = np.array([0, 0, G_NORM])
        self.imu_rec, self.wheel_rec = [], []
        self.imu_raw, self.wheel_raw = [], []
        for k in range(self.n_kf - 1):
            t0 = self.kf_t[k]
            for hz, store, raw in ((IMU_HZ, self.imu_rec, self.imu_raw), (WHEEL_HZ, self.wheel_rec, self.wheel_raw)):
                n = int(round(KF_DT * hz))
                dt = KF_DT / n
                rows = []
                for s in range(n + 1):
                    p, R, v, a, om = self.kinematics(t0 + s * dt)
                    if hz == IMU_HZ:
                        acc = R.T @ (a + G) + self.ba + sc * rng.normal(0, ACC_N, 3)
                        gyr = om + self.bg + sc * rng.normal(0, GYR_N, 3)
                        rows.append(np.concatenate([[dt], acc, gyr]))
                    else:
                        vo = self.rio.T @ (R.T @ v + np.cross(om, self.tio)) + sc * rng.normal(0, VEL_N_WHEEL, 3)
                        go = self.rio.T @ om + sc * rng.normal(0, GYR_N_WHEEL, 3)
                        rows.append(np.concatenate([[dt], vo, go]))
                rows = np.array(rows)
                first, samples = rows[0, 1:], rows[1:]
                raw.append((samples, first))
                if hz == IMU_HZ:
                    store.append(preintegrate_imu_np(samples, first, self.ba_est, self.bg_est))
                else:
                    store.append(preintegrate_wheel_np(samples, first, np.array([1.0, 1.0, 1.0, 0.0])))

    def truth_state(self, k0):
        """True parameter blocks of window k0 (dict, same keys as a snapshot's state part)."""
        pose, sb = np.zeros((abi.NFRAMES, 7)), np.zeros((abi.NFRAMES, 9))
        for i in range(abi.NFRAMES):
            p, R, v, _, _ = self.kinematics(self.kf_t[k0 + i])
            q = rot2q(R)
            pose[i] = np.concatenate([p, q / np.linalg.norm(q)])
            sb[i] = np.concatenate([v, self.ba, self.bg])
        qic, qio = rot2q(self.ric), rot2q(self.rio)
        return dict(pose=pose, speed_bias=sb, ex_pose=np.concatenate([self.tic, qic / np.linalg.norm(qic)]),
                    ex_pose_wheel=np.concatenate([self.tio, qio / np.linalg.norm(qio)]),
                    ix_wheel=np.ones(3), td=0.0, td_wheel=0.0)

    def _landmarks(self, k0, rng):
        """Tracks for window k0: start ~U{0..7}, length ~U{4..11-start}."""
        sc = 1.0 if self.noise else 0.0
        L = self.L
        poses = [self.pose(self.kf_t[k0 + i]) for i in range(abi.NFRAMES)]
        feats = []
        while len(feats) < L:
            start = int(rng.integers(0, 8))
            length = int(rng.integers(4, 11 - start + 1))
            depth = rng.uniform(1.0, 10.0)
            u, v = rng.uniform(0, 640), rng.uniform(0, 480)
            pn = np.array([(u - 320) / FOCAL, (v - 240) / FOCAL, 1.0])
            p, R = poses[start]
            pw = R @ (self.ric @ (pn * depth) + self.tic) + p
            obs, ok = [], True
            for j in range(start, start + length):
                pj, Rj = poses[j]
                pc = self.ric.T @ (Rj.T @ (pw - pj) - self.tic)
                if pc[2] < 0.5:
                    ok = False
                    break
                xy = pc[:2] / pc[2] + sc * rng.normal(0, 0.5 / FOCAL, 2)
                obs.append(xy)
            if not ok:
                continue
            obs = np.array(obs)
            vel = np.zeros_like(obs)
            vel[1:] = (obs[1:] - obs[:-1]) / KF_DT
            vel[0] = vel[1]
            inv_depth0 = 1.0 / depth
            feats.append(dict(start=start, obs=obs, vel=vel, inv_depth=inv_depth0))
        return feats

    def feature_list(self, k0, extra_short=0):
        """FeatureManager-style list (std::list<FeaturePerId> flattened) for window k0, plus
        `extra_short` features with < 4 observations interleaved (they must be skipped)."""
        rng = np.random.default_rng(self.seed + 1000 * (k0 + 1))
        sc = 1.0 if self.noise else 0.0
        feats = self._landmarks(k0, rng)
        for _ in range(extra_short):
            n = int(rng.integers(1, 4))
            pos = int(rng.integers(0, len(feats) + 1))
            feats.insert(pos, dict(start=int(rng.integers(0, 11 - n + 1)), obs=rng.normal(0, 0.3, (n, 2)),
                                   vel=np.zeros((n, 2)), inv_depth=0.2))
        start = np.array([f["start"] for f in feats], np.int32)
        n_obs = np.array([len(f["obs"]) for f in feats], np.int32)
        rows, tds = [], []
        for f in feats:
            for o, v in zip(f["obs"], f["vel"]):
                rows.append([o[0], o[1], 1.0, FOCAL * o[0] + 320, FOCAL * o[1] + 240, v[0], v[1]])
                tds.append(0.0)
        est = np.array([1.0 / (f["inv_depth"] * (1.0 + sc * rng.normal(0, 0.1))) for f in feats])
        return dict(start_frame=start, n_obs=n_obs, obs=np.array(rows), obs_td=np.array(tds),
                    estimated_depth=est, estimate_flag=np.zeros(len(feats), np.int32))

    def initial_state(self, k0):
        rng = np.random.default_rng(self.seed + 7 + 1000 * (k0 + 1))
        sc = 1.0 if self.noise else 0.0
        st = self.truth_state(k0)
        for i in range(abi.NFRAMES):
            st["pose"][i, :3] += sc * rng.normal(0, 0.02, 3)
            dq = so3_exp(sc * rng.normal(0, np.deg2rad(0.5), 3))
            q = qmul(st["pose"][i, 3:], dq)
            st["pose"][i, 3:] = q / np.linalg.norm(q)
            st["speed_bias"][i, :3] += sc * rng.normal(0, 0.05, 3)
            st["speed_bias"][i, 3:6] = self.ba_est
            st["speed_bias"][i, 6:9] = self.bg_est
        return st

    def window(self, k0=0, state=None, prior=None, factors=None):
        """Snapshot dict of window k0. `factors` = output of build_visual_factors (dict) or None to
        build the list here with numpy (same loop as estimator.cpp:3330-3358)."""
        st = state if state is not None else self.initial_state(k0)
        snap = dict(st)
        snap["frame_count"] = abi.WINDOW_SIZE
        if factors is None:
            factors = build_visual_factors_np(self.feature_list(k0))
        snap.update(factors)
        snap["imu"] = np.array(self.imu_rec[k0:k0 + 10])
        snap["imu_frame"] = np.arange(10, dtype=np.int32)
        if self.use_wheel:
            snap["wheel"] = np.array(self.wheel_rec[k0:k0 + 10])
            snap["wheel_frame"] = np.arange(10, dtype=np.int32)
        # SetParameterBlockConstant decisions of the shipped m3dgr.yaml: estimate_extrinsic 0 (:33),
        # estimate_wheel_extrinsic 1 (:64), estimate_wheel_intrinsic 0 (:125), estimate_td 0 (:137),
        # estimate_td_wheel 0 (:140)
        snap["ex_cam_const"] = 1
        snap["ex_wheel_const"] = 0 if (self.use_wheel and self.est_wheel_ex) else 1
        snap["ix_wheel_const"] = 1
        snap["td_const"] = 1
        snap["td_wheel_const"] = 1
        snap["prior"] = prior
        return snap


def build_visual_factors_np(fl):
    """numpy statement of the landmark bookkeeping (feature_manager.cpp:43-55,286-302;
    estimator.cpp:3330-3358): k-th feature with >= 4 observations <-> para_Feature[k]."""
    idx, ii, jj, pi, pj, vi, vj, tdi, tdj, lam, fconst = [], [], [], [], [], [], [], [], [], [], []
    off = 0
    k = -1
    for f in range(len(fl["n_obs"])):
        n, s = int(fl["n_obs"][f]), int(fl["start_frame"][f])
        rows = fl["obs"][off:off + n]
        tds = fl["obs_td"][off:off + n]
        off += n
        if n < 4:
            continue
        k += 1
        lam.append(1.0 / fl["estimated_depth"][f])
        fconst.append(1 if fl["estimate_flag"][f] == 1 else 0)
        for o in range(1, n):
            idx.append(k)
            ii.append(s)
            jj.append(s + o)
            pi.append(rows[0, :3])
            pj.append(rows[o, :3])
            vi.append(rows[0, 5:7])
            vj.append(rows[o, 5:7])
            tdi.append(tds[0])
            tdj.append(tds[o])
    return dict(vis_feature_index=np.array(idx, np.int32), vis_imu_i=np.array(ii, np.int32),
                vis_imu_j=np.array(jj, np.int32), vis_pts_i=np.array(pi).reshape(-1, 3),
                vis_pts_j=np.array(pj).reshape(-1, 3), vis_vel_i=np.array(vi).reshape(-1, 2),
                vis_vel_j=np.array(vj).reshape(-1, 2), vis_td_i=np.array(tdi, float), vis_td_j=np.array(tdj, float),
                para_feature=np.array(lam, float), feature_const=np.array(fconst, np.uint8))


def shift_state_for_next_window(scn, solved_state, k_next):
    """Initial state of window k_next from the solved state of window k_next-1: frames 1..10 move
    to slots 0..9 (slideWindow, estimator.cpp:3700-3760); the new frame comes from truth + noise."""
    st = scn.initial_state(k_next)
    st["pose"][:10] = solved_state["pose"][1:]
    st["speed_bias"][:10] = solved_state["speed_bias"][1:]
    for key in ("ex_pose", "ex_pose_wheel", "ix_wheel", "td", "td_wheel"):
        st[key] = solved_state[key]
    return st


# ------------------------------------------------------------------ BASELINE configs[3]: global_fusion pose graph
def pose_graph(n=5000, seed=20250708 + 4, fix_every=10, noise=True):
    """n poses on a planar random walk, one RelativeRTError per consecutive pair (the VIO
    odometry: sigma_t 0.1 * |step|, sigma_q 0.01 rad — weights t_var 0.1, q_var 0.01 as in globalOpt.cpp:171-173), a
    position fix TError every `fix_every`-th pose (sigma 0.5 m). Poses are [t(3), q(w,x,y,z)] (globalOpt.cpp:46-47).
    The initial guess is the dead-reckoned (drifting) chain, as the reference seeds globalPoseMap from the VIO poses."""
    rng = np.random.default_rng(seed)
    sc = 1.0 if noise else 0.0

    def q_wxyz(yaw, pitch=0.0, roll=0.0):
        q = rot2q(rz(yaw) @ ry(pitch) @ rx(roll))          # x y z w
        q = q / np.linalg.norm(q)
        return np.array([q[3], q[0], q[1], q[2]])

    def qm(a, b):
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])

    def R_of(q):
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    truth = np.zeros((n, 7))
    yaw, p = 0.0, np.zeros(3)
    for i in range(n):
        truth[i] = np.concatenate([p, q_wxyz(yaw, 0.01 * np.sin(0.01 * i), 0.01 * np.cos(0.013 * i))])
        yaw += rng.normal(0, 0.05)
        p = p + np.array([np.cos(yaw), np.sin(yaw), 0.0]) * 0.1 + np.array([0, 0, rng.normal(0, 0.002)])
    rel_i = np.arange(n - 1, dtype=np.int32)
    rel_meas = np.zeros((n - 1, 7))
    for k in range(n - 1):
        Ri = R_of(truth[k, 3:])
        t = Ri.T @ (truth[k + 1, :3] - truth[k, :3]) + sc * rng.normal(0, 0.002, 3)
        q = qm(truth[k, 3:] * np.array([1, -1, -1, -1]), truth[k + 1, 3:])
        dth = sc * rng.normal(0, 0.0005, 3)
        q = qm(q, np.concatenate([[1.0], 0.5 * dth]))
        rel_meas[k] = np.concatenate([t, q / np.linalg.norm(q)])
    fix_i = np.arange(0, n, fix_every, dtype=np.int32)
    fix_meas = np.column_stack([truth[fix_i, :3] + sc * rng.normal(0, 0.5, (len(fix_i), 3)), np.full(len(fix_i), 0.5)])
    init = np.zeros((n, 7))
    init[0] = truth[0]
    for k in range(n - 1):
        Ri = R_of(init[k, 3:])
        q = qm(init[k, 3:], rel_meas[k, 3:])
        init[k + 1] = np.concatenate([init[k, :3] + Ri @ rel_meas[k, :3], q / np.linalg.norm(q)])
    return dict(pose=init, truth=truth, rel_i=rel_i, rel_meas=rel_meas, fix_i=fix_i, fix_meas=fix_meas)


def lidar_block(scn, k0, n=2000, seed=0, noise=0.02, sqrt_info=20.0, huber_delta=0.5, frame=None, outliers=0.0):
    """BASELINE configs[4]: n point-to-plane factors of a LiDAR scan taken at the newest pose of window k0 (max_num_residuals:
    2000, lio/config/m3dgr.yaml:45): body-frame points at 2-20 m, a plane through every point's true world position with
    a random unit normal, range noise `noise` along the normal (a fraction `outliers` of them displaced by 1 m)."""
    frame = abi.WINDOW_SIZE if frame is None else frame
    rng = np.random.default_rng(seed + 31 * k0)
    p, R = scn.pose(scn.kf_t[k0 + frame])
    d = rng.normal(size=(n, 3))
    pts = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(2.0, 20.0, (n, 1))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pw = pts @ R.T + p
    err = rng.normal(0, noise, n) + np.where(rng.random(n) < outliers, 1.0, 0.0)
    offs = -(nrm * pw).sum(axis=1) + err
    return dict(frame=frame, pts=pts, normals=nrm, offsets=offs, weights=rng.uniform(0.5, 1.0, n), sqrt_info=sqrt_info,
                huber_delta=huber_delta)
