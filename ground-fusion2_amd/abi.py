"""ctypes mirror of include/gfbe.h (the C-ABI drop-in boundary of Estimator::optimization(),
reference: Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:2951-3698).

A *window snapshot* is a plain dict of numpy arrays holding everything one optimization() call
reads. `make_window` turns it into a `gfbe_window` whose pointers alias the
numpy buffers (the returned holder keeps them alive).
"""
import ctypes as C
import numpy as np

WINDOW_SIZE = 10
NFRAMES = 11
DENSE_DIM = 246
CORE_DIM = 187      # the tangent dims without the GNSS blocks
MAX_PRIOR_BLOCKS = 32
PRIOR_X0_CAP = NFRAMES * 16 + 32

OK, NO_CONVERGENCE, NUMERICAL_FAILURE, BAD_INPUT, DEVICE_ERROR, NO_DEVICE = range(6)
MARGIN_OLD, MARGIN_SECOND_NEW, MARGIN_NONE = 0, 1, 2

BLK_POSE0, BLK_SB0, BLK_EX_CAM, BLK_EX_WHEEL = 0, 11, 22, 23
BLK_SX, BLK_SY, BLK_SW, BLK_TD, BLK_TD_WHEEL, BLK_PLANE_R, BLK_PLANE_Z = 24, 25, 26, 27, 28, 29, 30
BLK_ANC_ECEF, BLK_YAW_ENU, BLK_RCV_DT0, BLK_RCV_DDT0, BLK_COUNT = 31, 32, 33, 77, 88     # GNSS blocks (rcv_dt: + 4 frame + constellation)

c_d = C.c_double
c_i = C.c_int32
c_u8 = C.c_uint8
PD = C.POINTER(c_d)
PI = C.POINTER(c_i)
PU8 = C.POINTER(c_u8)


def block_global_size(bid):
    if bid < BLK_SB0:
        return 7
    if bid < BLK_EX_CAM:
        return 9
    if bid in (BLK_EX_CAM, BLK_EX_WHEEL):
        return 7
    if bid == BLK_PLANE_R:
        return 4
    if bid == BLK_ANC_ECEF:
        return 3
    return 1


def block_tangent_offset(bid):
    """First tangent dim of block `bid` in the DENSE_DIM-wide layout of the solver."""
    if bid < BLK_SB0:
        return 6 * bid
    if bid < BLK_EX_CAM:
        return 73 + 9 * (bid - BLK_SB0)
    if bid >= BLK_RCV_DDT0:
        return 235 + (bid - BLK_RCV_DDT0)
    if bid >= BLK_RCV_DT0:
        return 191 + (bid - BLK_RCV_DT0)
    return {BLK_EX_CAM: 66, BLK_TD: 72, BLK_EX_WHEEL: 172, BLK_SX: 178, BLK_SY: 179, BLK_SW: 180, BLK_TD_WHEEL: 181, BLK_PLANE_R: 182,
            BLK_PLANE_Z: 186, BLK_ANC_ECEF: 187, BLK_YAW_ENU: 190}[bid]


def block_local_size(bid):
    g = block_global_size(bid)
    return 6 if g == 7 else g


class GnssObs(C.Structure):
    _fields_ = [("sv_pos", c_d * 3), ("sv_vel", c_d * 3), ("svdt", c_d), ("svddt", c_d), ("tgd", c_d), ("pr_uura", c_d), ("dp_uura", c_d),
                ("psr", c_d), ("dopp", c_d), ("wavelength", c_d), ("ratio", c_d), ("doy", c_d), ("tow", c_d),
                ("frame", c_i), ("lower_idx", c_i), ("sys_idx", c_i), ("_pad", c_i)]


class GnssState(C.Structure):
    _fields_ = [("rcv_dt", (c_d * 4) * NFRAMES), ("rcv_ddt", c_d * NFRAMES), ("yaw_enu_local", c_d), ("anc_ecef", c_d * 3)]


class State(C.Structure):
    _fields_ = [("para_Pose", (c_d * 7) * NFRAMES),
                ("para_SpeedBias", (c_d * 9) * NFRAMES),
                ("para_Ex_Pose", c_d * 7),
                ("para_Ex_Pose_wheel", c_d * 7),
                ("para_Ix_wheel", c_d * 3),
                ("para_Td", c_d),
                ("para_Td_wheel", c_d),
                ("para_plane_R", c_d * 4),
                ("para_plane_Z", c_d),
                ("gnss", GnssState)]


class ImuPreint(C.Structure):
    _fields_ = [("sum_dt", c_d), ("delta_p", c_d * 3), ("delta_q", c_d * 4), ("delta_v", c_d * 3),
                ("linearized_ba", c_d * 3), ("linearized_bg", c_d * 3),
                ("jacobian", c_d * 225), ("covariance", c_d * 225)]


class WheelPreint(C.Structure):
    _fields_ = [("sum_dt", c_d), ("delta_p", c_d * 3), ("delta_q", c_d * 4),
                ("linearized_sx", c_d), ("linearized_sy", c_d), ("linearized_sw", c_d), ("linearized_td", c_d),
                ("linearized_vel", c_d * 3), ("linearized_gyr", c_d * 3),
                ("vel_1", c_d * 3), ("gyr_1", c_d * 3),
                ("jacobian", c_d * 18), ("covariance", c_d * 36)]


IMU_DOUBLES = C.sizeof(ImuPreint) // 8      # 467
WHEEL_DOUBLES = C.sizeof(WheelPreint) // 8  # 78


class Prior(C.Structure):
    _fields_ = [("valid", c_i), ("n", c_i), ("n_blocks", c_i),
                ("block_id", c_i * MAX_PRIOR_BLOCKS), ("block_size", c_i * MAX_PRIOR_BLOCKS),
                ("block_idx", c_i * MAX_PRIOR_BLOCKS),
                ("x0", c_d * PRIOR_X0_CAP),
                ("J0", PD), ("r0", PD)]


class Visual(C.Structure):
    _fields_ = [("n_factor", c_i), ("feature_index", PI), ("imu_i", PI), ("imu_j", PI),
                ("pts_i", PD), ("pts_j", PD), ("vel_i", PD), ("vel_j", PD), ("td_i", PD), ("td_j", PD)]


class LioBlock(C.Structure):
    _fields_ = [("n", c_i), ("frame", c_i), ("pts", PD), ("normals", PD), ("offsets", PD), ("weights", PD),
                ("sqrt_info", c_d), ("huber_delta", c_d)]


class Window(C.Structure):
    _fields_ = [("frame_count", c_i), ("state", State),
                ("n_feature", c_i), ("para_Feature", PD), ("feature_const", PU8),
                ("pose_const", c_u8 * NFRAMES), ("sb_const", c_u8 * NFRAMES),
                ("ex_cam_const", c_u8), ("ex_wheel_const", c_u8), ("ix_wheel_const", c_u8),
                ("td_const", c_u8), ("td_wheel_const", c_u8),
                ("ex_cam_mask", c_u8 * 6), ("ex_wheel_mask", c_u8 * 6),
                ("use_plane", c_u8), ("plane_const", c_u8), ("use_anchor", c_u8),
                ("n_imu", c_i), ("imu_frame", PI), ("imu", C.POINTER(ImuPreint)),
                ("n_wheel", c_i), ("wheel_frame", PI), ("wheel", C.POINTER(WheelPreint)),
                ("vis", Visual),
                ("prior", C.POINTER(Prior)),
                ("lio", LioBlock),
                ("plane_noise_inv", c_d * 3), ("anchor_pose", c_d * 7), ("anchor_sqrt_info", c_d),
                ("gnss_ready", c_i), ("n_gnss", c_i), ("gnss_obs", C.POINTER(GnssObs)), ("gnss_iono", PD),
                ("gnss_frame_dt", c_d * WINDOW_SIZE), ("gnss_ddt_weight", c_d)]


class Options(C.Structure):
    _fields_ = [("struct_size", c_i), ("max_num_iterations", c_i), ("huber_delta", c_d), ("vis_sqrt_info", c_d), ("g_norm", c_d),
                ("initial_trust_region_radius", c_d), ("function_tolerance", c_d), ("gradient_tolerance", c_d),
                ("parameter_tolerance", c_d), ("min_relative_decrease", c_d), ("jacobi_scaling", c_i),
                ("marg_eps", c_d), ("marg_sqrt", c_i), ("use_graph", c_i), ("split_batch", c_i),
                ("max_solver_time_in_seconds", c_d), ("host_threads", c_i), ("solve_kernel", c_i),
                ("test_fail_chol_iter", c_i), ("test_fail_chol_count", c_i), ("sharded_mu_retries", c_i), ("speculative_linearization", c_i), ("merge_lin_schur", c_i)]


class Summary(C.Structure):
    _fields_ = [("status", c_i), ("iterations", c_i), ("num_successful", c_i), ("termination", c_i),
                ("initial_cost", c_d), ("final_cost", c_d), ("final_radius", c_d),
                ("cost_history", c_d * 16), ("accepted", c_u8 * 16),
                ("ms_solve", c_d), ("ms_marginalize", c_d), ("bytes_uploaded", c_d), ("bytes_downloaded", c_d)]


class FeatureList(C.Structure):
    _fields_ = [("n", c_i), ("start_frame", PI), ("n_obs", PI), ("obs_offset", PI),
                ("obs", PD), ("obs_td", PD), ("estimated_depth", PD), ("estimate_flag", PI)]


def default_options():
    """Solver options the reference runs with (estimator.cpp:193,2959,3364-3376; m3dgr.yaml:108-117)."""
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.max_num_iterations = 8
    o.huber_delta = 1.0
    o.vis_sqrt_info = 600.0 / 1.5
    o.g_norm = 9.7944
    o.initial_trust_region_radius = 1e4
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.min_relative_decrease = 1e-3
    o.jacobi_scaling = 1
    o.marg_eps = 1e-8
    o.marg_sqrt = 1
    o.use_graph = 0
    o.split_batch = 1
    o.max_solver_time_in_seconds = 0.0
    o.host_threads = 0
    o.solve_kernel = 0
    o.test_fail_chol_iter = 0
    o.test_fail_chol_count = 1
    o.sharded_mu_retries = 1
    o.speculative_linearization = 1
    o.merge_lin_schur = 0
    return o


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _pd(a):
    return a.ctypes.data_as(PD)


def _pi(a):
    return a.ctypes.data_as(PI)


def state_from_snapshot(snap, st=None):
    st = st or State()
    pose = _f64(snap["pose"]).reshape(NFRAMES, 7)
    sb = _f64(snap["speed_bias"]).reshape(NFRAMES, 9)
    for i in range(NFRAMES):
        st.para_Pose[i][:] = pose[i].tolist()
        st.para_SpeedBias[i][:] = sb[i].tolist()
    st.para_Ex_Pose[:] = _f64(snap["ex_pose"]).tolist()
    st.para_Ex_Pose_wheel[:] = _f64(snap["ex_pose_wheel"]).tolist()
    st.para_Ix_wheel[:] = _f64(snap["ix_wheel"]).tolist()
    st.para_Td = float(snap["td"])
    st.para_Td_wheel = float(snap["td_wheel"])
    st.para_plane_R[:] = _f64(snap.get("plane_R", [0.0, 0.0, 0.0, 1.0])).tolist()
    st.para_plane_Z = float(snap.get("plane_Z", 0.0))
    g = snap.get("gnss_state")
    if g is not None:
        rd = _f64(g["rcv_dt"]).reshape(NFRAMES, 4)
        for i in range(NFRAMES):
            st.gnss.rcv_dt[i][:] = rd[i].tolist()
        st.gnss.rcv_ddt[:] = _f64(g["rcv_ddt"]).tolist()
        st.gnss.yaw_enu_local = float(g["yaw_enu_local"])
        st.gnss.anc_ecef[:] = _f64(g["anc_ecef"]).tolist()
    return st


def state_to_dict(st):
    return {
        "pose": np.array([list(st.para_Pose[i]) for i in range(NFRAMES)]),
        "speed_bias": np.array([list(st.para_SpeedBias[i]) for i in range(NFRAMES)]),
        "ex_pose": np.array(list(st.para_Ex_Pose)),
        "ex_pose_wheel": np.array(list(st.para_Ex_Pose_wheel)),
        "ix_wheel": np.array(list(st.para_Ix_wheel)),
        "td": float(st.para_Td),
        "td_wheel": float(st.para_Td_wheel),
        "plane_R": np.array(list(st.para_plane_R)),
        "plane_Z": float(st.para_plane_Z),
        "gnss_state": {"rcv_dt": np.array([list(st.gnss.rcv_dt[i]) for i in range(NFRAMES)]), "rcv_ddt": np.array(list(st.gnss.rcv_ddt)),
                       "yaw_enu_local": float(st.gnss.yaw_enu_local), "anc_ecef": np.array(list(st.gnss.anc_ecef))},
    }


def flat_state(d):
    """A state dict with the GNSS blocks lifted to the top level (gnss_rcv_dt, ...): every value an array or a scalar."""
    out = {k: v for k, v in d.items() if k != "gnss_state"}
    for k, v in (d.get("gnss_state") or {}).items():
        out["gnss_" + k] = v
    return out


class PriorHolder:
    """Owns J0/r0 storage for a gfbe_prior (capacity DENSE_DIM) and converts to/from dicts."""

    def __init__(self, d=None):
        self.J0 = np.zeros(DENSE_DIM * DENSE_DIM)
        self.r0 = np.zeros(DENSE_DIM)
        self.c = Prior()
        self.c.J0 = _pd(self.J0)
        self.c.r0 = _pd(self.r0)
        self.c.valid = 0
        if d is not None:
            self.load(d)

    def load(self, d):
        n = int(d["n"])
        nb = len(d["block_id"])
        self.c.valid = int(d.get("valid", 1))
        self.c.n = n
        self.c.n_blocks = nb
        for k in range(nb):
            self.c.block_id[k] = int(d["block_id"][k])
            self.c.block_size[k] = int(d["block_size"][k])
            self.c.block_idx[k] = int(d["block_idx"][k])
        x0 = _f64(d["x0"]).ravel()
        for k in range(len(x0)):
            self.c.x0[k] = x0[k]
        self.J0[: n * n] = _f64(d["J0"]).ravel()
        self.r0[:n] = _f64(d["r0"]).ravel()

    def to_dict(self):
        n, nb = self.c.n, self.c.n_blocks
        sizes = [self.c.block_size[k] for k in range(nb)]
        return {"valid": int(self.c.valid), "n": n,
                "block_id": np.array([self.c.block_id[k] for k in range(nb)], dtype=np.int32),
                "block_size": np.array(sizes, dtype=np.int32),
                "block_idx": np.array([self.c.block_idx[k] for k in range(nb)], dtype=np.int32),
                "x0": np.array([self.c.x0[k] for k in range(sum(sizes))]),
                "J0": self.J0[: n * n].reshape(n, n).copy(), "r0": self.r0[:n].copy()}


class WindowHolder:
    """gfbe_window + the numpy buffers its pointers alias."""

    def __init__(self, snap):
        self.snap = snap
        w = Window()
        self.c = w
        w.frame_count = int(snap.get("frame_count", WINDOW_SIZE))
        state_from_snapshot(snap, w.state)
        self.lam = _f64(snap["para_feature"])
        L = self.lam.shape[0]
        self.fconst = _u8(snap.get("feature_const", np.zeros(L, np.uint8)))
        w.n_feature = L
        w.para_Feature = _pd(self.lam)
        w.feature_const = self.fconst.ctypes.data_as(PU8)
        w.pose_const[:] = _u8(snap.get("pose_const", np.zeros(NFRAMES))).tolist()
        w.sb_const[:] = _u8(snap.get("sb_const", np.zeros(NFRAMES))).tolist()
        w.ex_cam_const = int(snap.get("ex_cam_const", 1))
        w.ex_wheel_const = int(snap.get("ex_wheel_const", 0))
        w.ix_wheel_const = int(snap.get("ix_wheel_const", 1))
        w.td_const = int(snap.get("td_const", 1))
        w.td_wheel_const = int(snap.get("td_wheel_const", 1))
        w.ex_cam_mask[:] = _u8(snap.get("ex_cam_mask", np.zeros(6))).tolist()
        w.ex_wheel_mask[:] = _u8(snap.get("ex_wheel_mask", np.zeros(6))).tolist()
        # optional in-window factors: snap["plane"] = dict(noise_inv=[pitch, roll, zpw], const=0/1); snap["anchor"] = dict(pose=[7], sqrt_info=120)
        pl, an = snap.get("plane"), snap.get("anchor")
        if pl is not None:
            w.use_plane, w.plane_const = 1, int(pl.get("const", 0))
            w.plane_noise_inv[:] = _f64(pl["noise_inv"]).tolist()
        if an is not None:
            w.use_anchor = 1
            w.anchor_pose[:] = _f64(an["pose"]).tolist()
            w.anchor_sqrt_info = float(an.get("sqrt_info", 120.0))
        # GNSS inside the window: snap["gnss"] = dict(obs=[dicts as for gnss_eval], iono=[8] or None, frame_dt=[10], ddt_weight, ready=1);
        # the GNSS state blocks travel in snap["gnss_state"]
        gn = snap.get("gnss")
        if gn is not None:
            self.gnss_obs = gnss_obs_array(gn["obs"])
            self.gnss_iono = _f64(gn["iono"]) if gn.get("iono") is not None else None
            w.gnss_ready, w.n_gnss = int(gn.get("ready", 1)), len(gn["obs"])
            w.gnss_obs = self.gnss_obs
            if self.gnss_iono is not None:
                w.gnss_iono = _pd(self.gnss_iono)
            w.gnss_frame_dt[:] = _f64(gn["frame_dt"]).tolist()
            w.gnss_ddt_weight = float(gn["ddt_weight"])
        # IMU / wheel
        self.imu = _f64(snap.get("imu", np.zeros((0, IMU_DOUBLES)))).reshape(-1, IMU_DOUBLES)
        self.imu_frame = _i32(snap.get("imu_frame", np.zeros(0)))
        w.n_imu = self.imu.shape[0]
        w.imu_frame = _pi(self.imu_frame)
        w.imu = C.cast(self.imu.ctypes.data, C.POINTER(ImuPreint))
        self.wheel = _f64(snap.get("wheel", np.zeros((0, WHEEL_DOUBLES)))).reshape(-1, WHEEL_DOUBLES)
        self.wheel_frame = _i32(snap.get("wheel_frame", np.zeros(0)))
        w.n_wheel = self.wheel.shape[0]
        w.wheel_frame = _pi(self.wheel_frame)
        w.wheel = C.cast(self.wheel.ctypes.data, C.POINTER(WheelPreint))
        # visual
        self.v_idx = _i32(snap["vis_feature_index"])
        self.v_i = _i32(snap["vis_imu_i"])
        self.v_j = _i32(snap["vis_imu_j"])
        self.v_pi = _f64(snap["vis_pts_i"]).reshape(-1, 3)
        self.v_pj = _f64(snap["vis_pts_j"]).reshape(-1, 3)
        self.v_vi = _f64(snap["vis_vel_i"]).reshape(-1, 2)
        self.v_vj = _f64(snap["vis_vel_j"]).reshape(-1, 2)
        self.v_tdi = _f64(snap["vis_td_i"])
        self.v_tdj = _f64(snap["vis_td_j"])
        v = w.vis
        v.n_factor = self.v_idx.shape[0]
        v.feature_index, v.imu_i, v.imu_j = _pi(self.v_idx), _pi(self.v_i), _pi(self.v_j)
        v.pts_i, v.pts_j, v.vel_i, v.vel_j = _pd(self.v_pi), _pd(self.v_pj), _pd(self.v_vi), _pd(self.v_vj)
        v.td_i, v.td_j = _pd(self.v_tdi), _pd(self.v_tdj)
        # prior
        self.prior = None
        if snap.get("prior") is not None:
            self.prior = PriorHolder(snap["prior"])
            w.prior = C.pointer(self.prior.c)
        # LiDAR factors on one pose: snap["lio"] = dict(frame, pts, normals, offsets, weights=None, sqrt_info, huber_delta)
        lio = snap.get("lio")
        if lio is not None and len(lio["pts"]):
            self.lio_pts, self.lio_normals = _f64(lio["pts"]).reshape(-1, 3), _f64(lio["normals"]).reshape(-1, 3)
            self.lio_offsets = _f64(lio["offsets"])
            self.lio_weights = _f64(lio["weights"]) if lio.get("weights") is not None else None
            w.lio.n, w.lio.frame = len(self.lio_pts), int(lio.get("frame", w.frame_count))
            w.lio.pts, w.lio.normals, w.lio.offsets = _pd(self.lio_pts), _pd(self.lio_normals), _pd(self.lio_offsets)
            if self.lio_weights is not None:
                w.lio.weights = _pd(self.lio_weights)
            w.lio.sqrt_info, w.lio.huber_delta = float(lio.get("sqrt_info", 1.0)), float(lio.get("huber_delta", 0.5))

    @property
    def n_vis(self):
        return self.c.vis.n_factor

    @property
    def n_feature(self):
        return self.c.n_feature


def bind(lib, prefix):
    """Attach argtypes/restype for the entry points shared by the product (gfbe_) and oracle (gfo_)."""
    P = C.POINTER
    f = getattr(lib, prefix + "build_visual_factors")
    f.restype = c_i
    f.argtypes = [P(FeatureList), c_i, PI, PI, PI, PD, PD, PD, PD, PD, PD, PD, PU8]
    f = getattr(lib, prefix + "feature_count")
    f.restype = c_i
    f.argtypes = [P(FeatureList)]
    f = getattr(lib, prefix + "visual_factor_count")
    f.restype = c_i
    f.argtypes = [P(FeatureList), c_i]
    f = getattr(lib, prefix + "set_depth")
    f.restype = None
    f.argtypes = [P(FeatureList), PD, PD, PI]
    return lib


def make_feature_list(fl):
    """fl: dict(start_frame[n], n_obs[n], obs[sum,7], obs_td[sum], estimated_depth[n], estimate_flag[n])."""
    h = {}
    h["start_frame"] = _i32(fl["start_frame"])
    h["n_obs"] = _i32(fl["n_obs"])
    off = np.zeros(len(h["n_obs"]), np.int32)
    if len(off) > 1:
        off[1:] = np.cumsum(h["n_obs"])[:-1]
    h["obs_offset"] = off
    h["obs"] = _f64(fl["obs"]).reshape(-1, 7)
    h["obs_td"] = _f64(fl["obs_td"])
    h["estimated_depth"] = _f64(fl["estimated_depth"])
    h["estimate_flag"] = _i32(fl["estimate_flag"])
    c = FeatureList()
    c.n = len(h["n_obs"])
    c.start_frame, c.n_obs, c.obs_offset = _pi(h["start_frame"]), _pi(h["n_obs"]), _pi(h["obs_offset"])
    c.obs, c.obs_td, c.estimated_depth = _pd(h["obs"]), _pd(h["obs_td"]), _pd(h["estimated_depth"])
    c.estimate_flag = _pi(h["estimate_flag"])
    h["c"] = c
    return h


class CApi:
    """Pythonic calls shared by the product library (prefix gfbe_, first arg = gfbe_ctx*) and the
    CPU oracle (prefix gfo_, first arg = const gfbe_options*). Subclasses set .lib/.prefix/.head."""

    lib = None
    prefix = ""
    head = None          # ctypes object passed as the first argument of compute entry points

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def check(self, rc, what):
        if rc not in (OK, NO_CONVERGENCE):
            raise RuntimeError("%s%s failed with status %d" % (self.prefix, what, rc))
        return rc

    # ---- a13 bookkeeping
    def build_visual_factors(self, fl, only_start_frame0=False):
        h = make_feature_list(fl)
        c = C.byref(h["c"])
        L = self._fn("feature_count")(c)
        K = self._fn("visual_factor_count")(c, int(only_start_frame0))
        idx, ii, jj = np.zeros(K, np.int32), np.zeros(K, np.int32), np.zeros(K, np.int32)
        pi, pj = np.zeros((K, 3)), np.zeros((K, 3))
        vi, vj = np.zeros((K, 2)), np.zeros((K, 2))
        tdi, tdj = np.zeros(K), np.zeros(K)
        lam, fc = np.zeros(L), np.zeros(L, np.uint8)
        k = self._fn("build_visual_factors")(c, int(only_start_frame0), _pi(idx), _pi(ii), _pi(jj), _pd(pi), _pd(pj),
                                            _pd(vi), _pd(vj), _pd(tdi), _pd(tdj), _pd(lam),
                                            fc.ctypes.data_as(PU8))
        assert k == K
        return dict(vis_feature_index=idx, vis_imu_i=ii, vis_imu_j=jj, vis_pts_i=pi, vis_pts_j=pj,
                    vis_vel_i=vi, vis_vel_j=vj, vis_td_i=tdi, vis_td_j=tdj, para_feature=lam, feature_const=fc)

    def set_depth(self, fl, para_feature):
        h = make_feature_list(fl)
        est = h["estimated_depth"].copy()
        flag = np.zeros(len(est), np.int32)
        lam = _f64(para_feature)
        self._fn("set_depth")(C.byref(h["c"]), _pd(lam), _pd(est), _pi(flag))
        return est, flag

    # ---- factor evaluation (block-CSR out)
    def eval_factors(self, snap, robustify=False):
        wh = snap if isinstance(snap, WindowHolder) else WindowHolder(snap)
        K, ni, nw = wh.n_vis, wh.c.n_imu, wh.c.n_wheel
        out = dict(vis_r=np.zeros((K, 2)), vis_J=np.zeros((K, 2, 20)), imu_r=np.zeros((ni, 15)),
                   imu_J=np.zeros((ni, 15, 30)), wheel_r=np.zeros((nw, 6)), wheel_J=np.zeros((nw, 6, 22)))
        npr = wh.prior.c.n if wh.prior is not None else 0
        out["prior_r"] = np.zeros(npr)
        cost = c_d(0.0)
        f = self._fn("eval_factors")
        f.restype = c_i
        rc = f(self.head, C.byref(wh.c), int(robustify), _pd(out["vis_r"]), _pd(out["vis_J"]), _pd(out["imu_r"]),
               _pd(out["imu_J"]), _pd(out["wheel_r"]), _pd(out["wheel_J"]),
               _pd(out["prior_r"]) if npr else None, C.byref(cost))
        self.check(rc, "eval_factors")
        out["cost"] = cost.value
        return out

    # ---- pre-integration
    def preintegrate_imu(self, intervals, lin_ba, lin_bg, noise):
        """intervals: list of (samples[n,7], first[6])."""
        n = len(intervals)
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum([len(s) for s, _ in intervals])
        samples = _f64(np.concatenate([s for s, _ in intervals]))
        first = _f64(np.array([f for _, f in intervals]))
        lin = _f64(np.tile(np.concatenate([lin_ba, lin_bg]), (n, 1)))
        out = np.zeros((n, IMU_DOUBLES))
        nz = _f64(noise)
        f = self._fn("preintegrate_imu")
        f.restype = c_i
        args = [n, _pi(off), _pd(samples), _pd(first), _pd(lin), _pd(nz), C.cast(out.ctypes.data, C.POINTER(ImuPreint))]
        rc = f(*(([self.head] if self.prefix == "gfbe_" else []) + args))
        self.check(rc, "preintegrate_imu")
        return out

    def preintegrate_wheel(self, intervals, lin, noise):
        n = len(intervals)
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum([len(s) for s, _ in intervals])
        samples = _f64(np.concatenate([s for s, _ in intervals]))
        first = _f64(np.array([f for _, f in intervals]))
        linv = _f64(np.tile(np.asarray(lin, float), (n, 1)))
        out = np.zeros((n, WHEEL_DOUBLES))
        nz = _f64(noise)
        f = self._fn("preintegrate_wheel")
        f.restype = c_i
        args = [n, _pi(off), _pd(samples), _pd(first), _pd(linv), _pd(nz), C.cast(out.ctypes.data, C.POINTER(WheelPreint))]
        rc = f(*(([self.head] if self.prefix == "gfbe_" else []) + args))
        self.check(rc, "preintegrate_wheel")
        return out

    # ---- the whole optimization() call
    def solve(self, snap, margin_flag=MARGIN_NONE):
        wh = snap if isinstance(snap, WindowHolder) else WindowHolder(snap)
        st = State()
        feat = np.zeros(wh.n_feature)
        pr = PriorHolder()
        sm = Summary()
        f = self._fn("solve_window")
        f.restype = c_i
        rc = f(self.head, C.byref(wh.c), int(margin_flag), C.byref(st), _pd(feat), C.byref(pr.c), C.byref(sm))
        self.check(rc, "solve_window")
        return dict(state=state_to_dict(st), feature=feat, prior=pr.to_dict() if pr.c.valid else None,
                    summary=summary_to_dict(sm), perf=summary_perf(sm), status=rc)


def _solve_raw(self, wh, margin_flag=MARGIN_NONE):
    """solve_window into outputs kept on the holder: the bare C call (latency measurements)."""
    if not hasattr(wh, "_raw_out"):
        wh._raw_out = (State(), np.zeros(max(wh.n_feature, 1)), PriorHolder(), Summary())
    st, feat, pr, sm = wh._raw_out
    f = self._fn("solve_window")
    f.restype = c_i
    return self.check(f(self.head, C.byref(wh.c), int(margin_flag), C.byref(st), _pd(feat), C.byref(pr.c), C.byref(sm)), "solve_window")


CApi.solve_raw = _solve_raw


def summary_to_dict(sm):
    n = sm.iterations + 1
    return dict(status=sm.status, iterations=sm.iterations, num_successful=sm.num_successful,
                termination=sm.termination, initial_cost=sm.initial_cost, final_cost=sm.final_cost,
                final_radius=sm.final_radius, cost_history=list(sm.cost_history)[:n],
                accepted=list(sm.accepted)[:n])


def summary_perf(sm):
    """The measured part of gfbe_summary (device phase times, PCIe bytes): kept apart from summary_to_dict, whose dicts the
    tests compare for bit-identity between runs."""
    return dict(ms_solve=sm.ms_solve, ms_marginalize=sm.ms_marginalize, bytes_uploaded=sm.bytes_uploaded,
                bytes_downloaded=sm.bytes_downloaded)


# ---------------------------------------------------------------------------------------------
# f1: feature tables (FeatureManager / slideWindow operations), shared by gfbe_ (device) and gfo_ (oracle)
# ---------------------------------------------------------------------------------------------
class FtabOptions(C.Structure):
    _fields_ = [("init_depth", c_d), ("min_parallax", c_d), ("focal_length", c_d), ("depth_threshold", c_d)]


def pose_rows(pose7):
    """[p, q(xyzw)] rows -> [P(3) | R(9, row-major)] rows, the pose argument of the feature-table calls."""
    pose7 = np.asarray(pose7, float).reshape(-1, 7)
    out = np.zeros((len(pose7), 12))
    for k, r in enumerate(pose7):
        x, y, z, w = r[3:] / np.linalg.norm(r[3:])
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        out[k, :3] = r[:3]
        out[k, 3:] = R.ravel()
    return out


class FeatureTables:
    """W feature tables behind `lib` (prefix gfbe_: device-resident, ctx = gfbe_ctx*; prefix gfo_: the CPU oracle,
    ctx ignored). Per-table arguments are lists of length W."""

    def __init__(self, lib, prefix, ctx, n_tables=1, capacity=4096, options=None):
        self.lib, self.prefix, self.ctx, self.W, self.cap = lib, prefix, ctx, n_tables, capacity
        self.h = C.c_void_p()
        for name in ("create", "add_frame", "remove_back_shift_depth", "remove_back", "remove_front", "remove_outlier",
                     "remove_failures", "clear_depth", "set_depth", "get_depth_vector", "triangulate", "check_outliers",
                     "size", "download"):
            self._f(name).restype = c_i
        self._f("destroy").restype = None
        opt = None
        if options is not None:
            opt = FtabOptions()
            self._f("default_options")(C.byref(opt))
            for k, v in options.items():
                setattr(opt, k, v)
        self._check(self._f("create")(self.ctx, int(n_tables), int(capacity), C.byref(opt) if opt is not None else None,
                                      C.byref(self.h)), "create")

    def _f(self, name):
        return getattr(self.lib, self.prefix + "ftab_" + name)

    def _check(self, rc, what):
        if rc != OK:
            raise RuntimeError("%sftab_%s failed with status %d" % (self.prefix, what, rc))

    def close(self):
        if self.h:
            self._f("destroy")(self.ctx, self.h)
            self.h = C.c_void_p()

    @staticmethod
    def _ragged(rows, dtype, width=None):
        off = np.zeros(len(rows) + 1, np.int32)
        off[1:] = np.cumsum([len(r) for r in rows])
        flat = np.concatenate([np.asarray(r, dtype).reshape(len(r), -1) if width else np.asarray(r, dtype).ravel() for r in rows]) \
            if off[-1] else np.zeros((0, width) if width else 0, dtype)
        return off, np.ascontiguousarray(flat)

    def add_frame(self, frame_count, ids, obs8, td):
        """ids[w]: ascending feature ids; obs8[w]: [n, 8]. Returns (keyframe[W], counters[W, 3], avg_parallax[W])."""
        off, fid = self._ragged(ids, np.int32)
        _, ob = self._ragged(obs8, np.float64, 8)
        fc, tdv = _i32(frame_count), _f64(td)
        kf, cnt, avg = np.zeros(self.W, np.int32), np.zeros((self.W, 3), np.int32), np.zeros(self.W)
        self._check(self._f("add_frame")(self.ctx, self.h, _pi(fc), _pi(off), _pi(fid), _pd(ob), _pd(tdv), _pi(kf), _pi(cnt), _pd(avg)), "add_frame")
        return kf, cnt, avg

    def remove_back_shift_depth(self, marg_pr, new_pr):
        a, b = _f64(marg_pr).reshape(self.W, 12), _f64(new_pr).reshape(self.W, 12)
        self._check(self._f("remove_back_shift_depth")(self.ctx, self.h, _pd(a), _pd(b)), "remove_back_shift_depth")

    def remove_back(self):
        self._check(self._f("remove_back")(self.ctx, self.h), "remove_back")

    def remove_front(self, frame_count):
        fc = _i32(frame_count)
        self._check(self._f("remove_front")(self.ctx, self.h, _pi(fc)), "remove_front")

    def remove_outlier(self, ids):
        off, flat = self._ragged(ids, np.int32)
        self._check(self._f("remove_outlier")(self.ctx, self.h, _pi(off), _pi(flat)), "remove_outlier")

    def remove_failures(self):
        self._check(self._f("remove_failures")(self.ctx, self.h), "remove_failures")

    def clear_depth(self):
        self._check(self._f("clear_depth")(self.ctx, self.h), "clear_depth")

    def set_depth(self, x):
        off, flat = self._ragged(x, np.float64)
        self._check(self._f("set_depth")(self.ctx, self.h, _pi(off), _pd(flat)), "set_depth")

    def get_depth_vector(self):
        off = (np.arange(self.W + 1) * self.cap).astype(np.int32)
        x, cnt = np.zeros(self.W * self.cap), np.zeros(self.W, np.int32)
        self._check(self._f("get_depth_vector")(self.ctx, self.h, _pi(off), _pd(x), _pi(cnt)), "get_depth_vector")
        return [x[off[w]:off[w] + cnt[w]].copy() for w in range(self.W)]

    def triangulate(self, poses, tic_ric, with_depth=False):
        p, e = _f64(poses).reshape(self.W, 11 * 12), _f64(tic_ric).reshape(self.W, 12)
        self._check(self._f("triangulate")(self.ctx, self.h, _pd(p), _pd(e), int(with_depth)), "triangulate")

    def check_outliers(self, poses, tic_ric, mode):
        p, e = _f64(poses).reshape(self.W, 11 * 12), _f64(tic_ric).reshape(self.W, 12)
        off = (np.arange(self.W + 1) * self.cap).astype(np.int32)
        ids, cnt = np.full(self.W * self.cap, -1, np.int32), np.zeros(self.W, np.int32)   # (touched pages: cheap pageable D2H)
        self._check(self._f("check_outliers")(self.ctx, self.h, _pd(p), _pd(e), int(mode), _pi(off), _pi(ids), _pi(cnt)), "check_outliers")
        return [ids[off[w]:off[w] + cnt[w]].copy() for w in range(self.W)]

    def size(self):
        n = np.zeros(self.W, np.int32)
        self._check(self._f("size")(self.ctx, self.h, _pi(n)), "size")
        return n

    def download(self, w=0):
        n = int(self.size()[w])
        out = dict(feature_id=np.zeros(n, np.int32), start_frame=np.zeros(n, np.int32), n_obs=np.zeros(n, np.int32),
                   obs8=np.zeros((n, 11, 8)), obs_td=np.zeros((n, 11)), estimated_depth=np.zeros(n),
                   estimate_flag=np.zeros(n, np.int32), solve_flag=np.zeros(n, np.int32))
        self._check(self._f("download")(self.ctx, self.h, int(w), _pi(out["feature_id"]), _pi(out["start_frame"]), _pi(out["n_obs"]),
                                        _pd(out["obs8"]), _pd(out["obs_td"]), _pd(out["estimated_depth"]), _pi(out["estimate_flag"]),
                                        _pi(out["solve_flag"])), "download")
        return out


def ftab_to_feature_list(tab):
    """FeatureTables.download() -> the flattened list gfbe_build_visual_factors consumes."""
    rows, tds = [], []
    for k in range(len(tab["n_obs"])):
        for o in range(tab["n_obs"][k]):
            rows.append(tab["obs8"][k, o, :7])
            tds.append(tab["obs_td"][k, o])
    return dict(start_frame=tab["start_frame"], n_obs=tab["n_obs"], obs=np.array(rows).reshape(-1, 7), obs_td=np.array(tds),
                estimated_depth=tab["estimated_depth"], estimate_flag=tab["estimate_flag"])


# ---------------------------------------------------------------------------------------------
# f3: global_fusion pose graph (gfbe_pg_* on the device, gfo_pg_* in the oracle)
# ---------------------------------------------------------------------------------------------
class PoseGraph:
    """Chain pose graph: poses [n,7] = t(3) q(wxyz); rel_i [m]; rel_meas [m,7]; fix_i [k]; fix_meas [k,4] = x y z var."""

    def __init__(self, lib, prefix, ctx):
        self.lib, self.prefix, self.ctx = lib, prefix, ctx
        for name in ("pg_eval", "pg_solve"):
            f = getattr(lib, prefix + name)
            f.restype = c_i
        getattr(lib, prefix + "pg_eval").argtypes = [C.c_void_p, c_i, PD, c_i, PI, PD, c_d, c_d, c_i, PI, PD, c_d, PD, PD, PD, PD]
        getattr(lib, prefix + "pg_solve").argtypes = [C.c_void_p, c_i, PD, c_i, PI, PD, c_d, c_d, c_i, PI, PD, c_d, c_i, PD, C.POINTER(Summary)]

    def _args(self, g):
        pose = _f64(g["pose"]).reshape(-1, 7)
        ri, rm = _i32(g["rel_i"]), _f64(g["rel_meas"]).reshape(-1, 7)
        fi, fmm = _i32(g["fix_i"]), _f64(g["fix_meas"]).reshape(-1, 4)
        return pose, ri, rm, fi, fmm

    def eval(self, g, t_var=0.1, q_var=0.01, huber=1.0):
        pose, ri, rm, fi, fmm = self._args(g)
        r, J, fr, cost = np.zeros((len(ri), 6)), np.zeros((len(ri), 6, 12)), np.zeros((len(fi), 3)), np.zeros(1)
        rc = getattr(self.lib, self.prefix + "pg_eval")(self.ctx, len(pose), _pd(pose), len(ri), _pi(ri), _pd(rm), t_var, q_var, len(fi),
                                                        _pi(fi), _pd(fmm), huber, _pd(r), _pd(J), _pd(fr), _pd(cost))
        if rc != OK:
            raise RuntimeError("%spg_eval failed with status %d" % (self.prefix, rc))
        return dict(rel_r=r, rel_J=J, fix_r=fr, cost=float(cost[0]))

    def solve(self, g, t_var=0.1, q_var=0.01, huber=1.0, max_iterations=5):
        pose, ri, rm, fi, fmm = self._args(g)
        out, sm = np.zeros_like(pose), Summary()
        rc = getattr(self.lib, self.prefix + "pg_solve")(self.ctx, len(pose), _pd(pose), len(ri), _pi(ri), _pd(rm), t_var, q_var, len(fi),
                                                         _pi(fi), _pd(fmm), huber, int(max_iterations), _pd(out), C.byref(sm))
        if rc not in (OK, NO_CONVERGENCE):
            raise RuntimeError("%spg_solve failed with status %d" % (self.prefix, rc))
        return dict(pose=out, summary=summary_to_dict(sm), status=rc)


def pg_plus(x, d6):
    """ceres::QuaternionParameterization::Plus on q (w,x,y,z) + identity on t; d6 = [dq(3), dt(3)]."""
    x, d6 = np.asarray(x, float), np.asarray(d6, float)
    nrm = np.linalg.norm(d6[:3])
    dq = np.concatenate([[np.cos(nrm)], np.sin(nrm) / nrm * d6[:3]]) if nrm > 0 else np.array([1.0, 0, 0, 0])
    a, b = dq, x[3:]
    q = np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                  a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])
    return np.concatenate([x[:3] + d6[3:], q])


# ---------------------------------------------------------------------------------------------
# f4: LIO point-to-plane factors (gfbe_lio_linearize / gfo_lio_linearize)
# ---------------------------------------------------------------------------------------------
def lio_linearize(lib, prefix, ctx, ct, pts, normals, offsets, alpha, weights, sqrt_info, pose_begin, pose_end=None, blocks=True):
    """blocks = False: only the normal equations (H = J^T J, g = J^T r, cost) come back — what a caller that feeds a solver needs;
    the per-factor residuals / Jacobians (r [n], J [n][6 or 12]) are not produced or copied."""
    pts, normals, offsets = _f64(pts).reshape(-1, 3), _f64(normals).reshape(-1, 3), _f64(offsets)
    n, dn = len(pts), 12 if ct else 6
    al = _f64(alpha if alpha is not None else np.zeros(n))
    wg = _f64(weights if weights is not None else np.ones(n))
    pb = _f64(pose_begin)
    pe = _f64(pose_end if pose_end is not None else pose_begin)
    r, J = (np.zeros(n), np.zeros((n, dn))) if blocks else (None, None)
    H, g, cost = np.zeros((dn, dn)), np.zeros(dn), np.zeros(1)
    f = getattr(lib, prefix + "lio_linearize")
    f.restype = c_i
    f.argtypes = [C.c_void_p, c_i, c_i, PD, PD, PD, PD, PD, c_d, PD, PD, PD, PD, PD, PD, PD]
    rc = f(ctx, int(ct), n, _pd(pts), _pd(normals), _pd(offsets), _pd(al), _pd(wg), float(sqrt_info), _pd(pb), _pd(pe),
           _pd(r) if blocks else None, _pd(J) if blocks else None, _pd(H), _pd(g), _pd(cost))
    if rc != OK:
        raise RuntimeError("%slio_linearize failed with status %d" % (prefix, rc))
    return dict(r=r, J=J, H=H, g=g, cost=float(cost[0]))


# ---------------------------------------------------------------------------------------------
# f2: optional in-window factors, evaluation only (gfbe_plane_eval / gfbe_anchor_eval / gfbe_orientation_subset_plus)
# ---------------------------------------------------------------------------------------------
def plane_eval(lib, prefix, ctx, pose, ex_wheel, plane_R, plane_Z, noise_inv):
    pose = _f64(pose).reshape(-1, 7)
    n = len(pose)
    ex, q, ni = _f64(ex_wheel), _f64(plane_R), _f64(noise_inv)
    r, J, cost = np.zeros((n, 3)), np.zeros((n, 3, 16)), np.zeros(1)
    f = getattr(lib, prefix + "plane_eval")
    f.restype = c_i
    f.argtypes = [C.c_void_p, c_i, PD, PD, PD, c_d, PD, PD, PD, PD]
    rc = f(ctx, n, _pd(pose), _pd(ex), _pd(q), float(plane_Z), _pd(ni), _pd(r), _pd(J), _pd(cost))
    if rc != OK:
        raise RuntimeError("%splane_eval failed with status %d" % (prefix, rc))
    return dict(r=r, J=J, cost=float(cost[0]))


def anchor_eval(lib, prefix, ctx, pose, anchor, sqrt_info=120.0):
    pose, anchor = _f64(pose).reshape(-1, 7), _f64(anchor).reshape(-1, 7)
    n = len(pose)
    r, J, cost = np.zeros((n, 6)), np.zeros((n, 6, 6)), np.zeros(1)
    f = getattr(lib, prefix + "anchor_eval")
    f.restype = c_i
    f.argtypes = [C.c_void_p, c_i, PD, PD, c_d, PD, PD, PD]
    rc = f(ctx, n, _pd(pose), _pd(anchor), float(sqrt_info), _pd(r), _pd(J), _pd(cost))
    if rc != OK:
        raise RuntimeError("%sanchor_eval failed with status %d" % (prefix, rc))
    return dict(r=r, J=J, cost=float(cost[0]))


def orientation_subset_plus(lib, prefix, q, delta, constant=(0, 0, 1)):
    q, d, m, out = _f64(q), _f64(delta), _u8(constant), np.zeros(4)
    f = getattr(lib, prefix + "orientation_subset_plus")
    f.restype = None
    f.argtypes = [PD, PD, PU8, PD]
    f(_pd(q), _pd(d), m.ctypes.data_as(PU8), _pd(out))
    return out


# ---------------------------------------------------------------------------------------------
# f2: GNSS factors, evaluation only (gfbe_gnss_eval)
# ---------------------------------------------------------------------------------------------
GNSS_OBS_KEYS = ("svdt", "svddt", "tgd", "pr_uura", "dp_uura", "psr", "dopp", "wavelength", "ratio", "doy", "tow", "frame", "lower_idx", "sys_idx")


def gnss_obs_array(obs):
    """list of dicts with sv_pos, sv_vel and GNSS_OBS_KEYS -> ctypes array of gfbe_gnss_obs"""
    arr = (GnssObs * max(len(obs), 1))()
    for k, o in enumerate(obs):
        arr[k].sv_pos[:] = [float(x) for x in o["sv_pos"]]
        arr[k].sv_vel[:] = [float(x) for x in o["sv_vel"]]
        for key in GNSS_OBS_KEYS:
            setattr(arr[k], key, o[key])
    return arr


def gnss_eval(lib, prefix, ctx, obs, iono, pose, speed_bias, rcv_dt, rcv_ddt, yaw_enu_local, anc_ecef, frame_dt, ddt_weight, want_J=True):
    """obs: list of dicts with sv_pos, sv_vel and GNSS_OBS_KEYS. pose [11][7], speed_bias [11][9] (the window state's blocks)."""
    n = len(obs)
    arr = gnss_obs_array(obs)
    st = State()
    p, sb = _f64(pose).reshape(NFRAMES, 7), _f64(speed_bias).reshape(NFRAMES, 9)
    for i in range(NFRAMES):
        st.para_Pose[i][:] = p[i].tolist()
        st.para_SpeedBias[i][:] = sb[i].tolist()
    g = GnssState()
    rd = _f64(rcv_dt).reshape(NFRAMES, 4)
    for i in range(NFRAMES):
        g.rcv_dt[i][:] = rd[i].tolist()
    g.rcv_ddt[:] = _f64(rcv_ddt).tolist()
    g.yaw_enu_local = float(yaw_enu_local)
    g.anc_ecef[:] = _f64(anc_ecef).tolist()
    fdt = _f64(frame_dt)
    assert fdt.shape == (WINDOW_SIZE,)
    io = _f64(iono) if iono is not None else None
    r, J, rc_, rs, cost = np.zeros((n, 2)), np.zeros((n, 2, 18)), np.zeros((4, WINDOW_SIZE)), np.zeros(WINDOW_SIZE), np.zeros(1)
    f = getattr(lib, prefix + "gnss_eval")
    f.restype = c_i
    f.argtypes = [C.c_void_p, c_i, C.POINTER(GnssObs), PD, C.POINTER(State), C.POINTER(GnssState), PD, c_d, PD, PD, PD, PD, PD]
    rc = f(ctx, n, arr, _pd(io) if io is not None else None, C.byref(st), C.byref(g), _pd(fdt), float(ddt_weight), _pd(r),
           _pd(J) if want_J else None, _pd(rc_), _pd(rs), _pd(cost))
    if rc != OK:
        raise RuntimeError("%sgnss_eval failed with status %d" % (prefix, rc))
    return dict(r=r, J=J if want_J else None, r_dt_ddt=rc_, r_smooth=rs, cost=float(cost[0]))
