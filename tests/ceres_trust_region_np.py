"""TEST INFRASTRUCTURE. An independent numpy statement of what ceres::Solve does with the reference's options
(estimator.cpp:3364-3379: DENSE_SCHUR, DOGLEG, everything else at Ceres 1.14's defaults), written from the PUBLISHED
algorithm — Ceres Solver 1.14 documentation, "Trust Region Methods" / "Dogleg" / Solver::Options — and NOT from
oracle/gfo_solver.cpp. It exists to remove the single-author risk on the one component of the path whose source is not in
/root/reference (a12): the oracle's trust-region loop must reproduce this one's accept / reject
sequence, costs and radii.

Deliberately different in structure from the oracle (and from the HIP kernels):
  * works on the stacked residual vector r and the full sparse-as-dense Jacobian J over ALL tangent dims (dense block AND
    landmarks) — no Schur complement, no normal-equation partials: the Gauss-Newton system (J^T J + mu D^2) y = -J^T r is
    factorised whole with numpy's Cholesky (mathematically what DENSE_SCHUR computes);
  * Jacobi scaling, the trust-region diagonal, the Cauchy point and the model cost change are computed from J and r
    (|J v|^2, not v^T H v);
  * own manifold Plus (pose_local_parameterization.cpp:12-36, pose_subset_parameterization.cpp:27-45).
What it shares with the oracle: residual blocks and their tangent Jacobians at a given point (`oracle.eval_factors`, which
tests/test_oracle_numpy.py pins against independent numpy formulas and central differences) and the robustified cost.

Restated algorithm (Ceres 1.14, trust_region_minimizer.cc / dogleg_strategy.cc as documented):
  iteration 0: evaluate; jacobi scaling s_j = 1 / (1 + |J_j|), fixed for the whole solve.
  every iteration: stop on max_num_iterations, gradient max-norm <= gradient_tolerance, radius < min_trust_region_radius.
    DoglegStrategy::ComputeStep (TRADITIONAL_DOGLEG), in the scaled coordinates J <- J diag(s):
      D = sqrt(clamp(diag(J^T J), min_lm_diagonal 1e-6, max_lm_diagonal 1e32)); everything below in x~ = D x
      g~ = J^T r / D;  alpha = |g~|^2 / |J (g~ / D)|^2            (Cauchy step -alpha g~)
      Gauss-Newton: (J^T J + mu D^2) y = -J^T r, mu from the strategy's state (min_mu 1e-8), x10 up to max_mu 1 while the
        linear solver fails;  gn~ = D y
      |gn~| <= radius: step~ = gn~;  alpha |g~| >= radius: step~ = -(radius / |g~|) g~;  otherwise the point of the segment
        Cauchy -> Gauss-Newton on the sphere (the numerically careful beta of dogleg_strategy.cc)
      step = step~ / D
    model_cost_change = -(J step) . (r + J step / 2); <= 0: invalid step (mu x10, recompute; 5 in a row: failure)
    candidate x (+) s step; parameter tolerance |x_c - x| <= eps (|x| + eps); function tolerance |dcost| <= eps cost
    rho = dcost / model_cost_change > min_relative_decrease (1e-3): accept — radius x 1/2 if rho < 1/4, radius =
      max(radius, 3 |step~|) if rho > 3/4, mu = max(min_mu, 2 mu / 10); else reject — radius x 1/2, same linearisation.
"""
import numpy as np

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth

MIN_DIAG, MAX_DIAG = 1e-6, 1e32
MIN_MU, MAX_MU, MU_INC = 1e-8, 1.0, 10.0
MIN_RADIUS = 1e-32


def _block_table(snap):
    """Free parameter blocks of the reduced program, in an order of this file's own: (name, index, global size, local size)."""
    fc = int(snap.get("frame_count", abi.WINDOW_SIZE))
    pose_const = np.asarray(snap.get("pose_const", np.zeros(abi.NFRAMES)), bool)
    sb_const = np.asarray(snap.get("sb_const", np.zeros(abi.NFRAMES)), bool)
    n_imu, n_wheel = len(snap.get("imu_frame", [])), len(snap.get("wheel_frame", []))
    K = len(snap["vis_imu_i"])
    prior = snap.get("prior")
    touched = set()
    for f in snap.get("imu_frame", []):
        touched |= {("pose", int(f)), ("pose", int(f) + 1), ("sb", int(f)), ("sb", int(f) + 1)}
    for f in snap.get("wheel_frame", []):
        touched |= {("pose", int(f)), ("pose", int(f) + 1), ("exw", 0), ("sx", 0), ("sy", 0), ("sw", 0), ("tdw", 0)}
    for k in range(K):
        touched |= {("pose", int(snap["vis_imu_i"][k])), ("pose", int(snap["vis_imu_j"][k])), ("exc", 0), ("td", 0)}
    if snap.get("plane") is not None and fc > 0:      # PlaneFactor on every pose i < frame_count (estimator.cpp:3214-3220)
        touched |= {("pose", i) for i in range(fc)} | {("exw", 0), ("plr", 0), ("plz", 0)}
    if snap.get("anchor") is not None:
        touched.add(("pose", 0))
    if gnss_factors_on(snap):            # estimator.cpp:3239-3291
        for o in snap["gnss"]["obs"]:
            lo = int(o["lower_idx"])
            touched |= {("pose", lo), ("sb", lo), ("pose", lo + 1), ("sb", lo + 1), ("dt", 4 * int(o["frame"]) + int(o["sys_idx"])), ("ddt", int(o["frame"])),
                        ("yaw", 0), ("anc", 0)}
        touched |= {("dt", q) for q in range(4 * abi.NFRAMES)} | {("ddt", i) for i in range(abi.NFRAMES)}
    if prior is not None:
        for bid in prior["block_id"]:
            touched.add(_block_of_id(int(bid)))
    const = set()
    for i in range(abi.NFRAMES):
        if pose_const[i] or i > fc:
            const.add(("pose", i))
        if sb_const[i] or i > fc:
            const.add(("sb", i))
    if int(snap.get("ex_cam_const", 1)):
        const.add(("exc", 0))
    if int(snap.get("ex_wheel_const", 0)):
        const.add(("exw", 0))
    if int(snap.get("td_const", 1)):
        const.add(("td", 0))
    if int(snap.get("td_wheel_const", 1)):
        const.add(("tdw", 0))
    if int(snap.get("ix_wheel_const", 1)):
        const |= {("sx", 0), ("sy", 0), ("sw", 0)}
    if snap.get("plane") is not None and int(snap["plane"].get("const", 0)):
        const |= {("plr", 0), ("plz", 0)}
    if snap.get("gnss") is not None and int(snap["gnss"].get("ready", 1)):
        const.add(("yaw", 0))            # estimator.cpp:2991
    order = [("ddt", i) for i in range(abi.NFRAMES)][::-1] + [("anc", 0)] + [("dt", q) for q in range(4 * abi.NFRAMES)] + [("yaw", 0)] + \
            [("plz", 0), ("exc", 0), ("td", 0)] + [("sb", i) for i in range(abi.NFRAMES)] + [("tdw", 0), ("sw", 0), ("sy", 0), ("sx", 0), ("exw", 0)] + \
            [("pose", i) for i in range(abi.NFRAMES)][::-1] + [("plr", 0)]
    # (global size, tangent size in the solve). para_plane_R: a quaternion on OrientationSubsetParameterization — 3 tangent dims
    sizes = {"pose": (7, 6), "sb": (9, 9), "exc": (7, 6), "exw": (7, 6), "td": (1, 1), "tdw": (1, 1), "sx": (1, 1), "sy": (1, 1), "sw": (1, 1),
             "plr": (4, 3), "plz": (1, 1), "anc": (3, 3), "yaw": (1, 1), "dt": (1, 1), "ddt": (1, 1)}
    free = [b for b in order if b in touched and b not in const]
    off, table = 0, {}
    for b in free:
        table[b] = off
        off += sizes[b[0]][1]
    return free, table, off, sizes


def gnss_factors_on(snap):
    """gnss_ready && !lowspeed (estimator.cpp:2965-2984, 3239): the mean absolute horizontal velocity over the window, as a vector."""
    gn = snap.get("gnss")
    if gn is None or not int(gn.get("ready", 1)):
        return False
    v = np.abs(np.asarray(snap["speed_bias"], float)[:, :2]).sum(axis=0) / abi.NFRAMES
    return not np.linalg.norm(v) < 0.3


def _block_of_id(bid):
    if bid >= abi.BLK_RCV_DDT0:
        return ("ddt", bid - abi.BLK_RCV_DDT0)
    if bid >= abi.BLK_RCV_DT0:
        return ("dt", bid - abi.BLK_RCV_DT0)
    if bid in (abi.BLK_ANC_ECEF, abi.BLK_YAW_ENU):
        return ("anc", 0) if bid == abi.BLK_ANC_ECEF else ("yaw", 0)
    if bid < abi.BLK_SB0:
        return ("pose", bid)
    if bid < abi.BLK_EX_CAM:
        return ("sb", bid - abi.BLK_SB0)
    return {abi.BLK_EX_CAM: ("exc", 0), abi.BLK_EX_WHEEL: ("exw", 0), abi.BLK_SX: ("sx", 0), abi.BLK_SY: ("sy", 0),
            abi.BLK_SW: ("sw", 0), abi.BLK_TD: ("td", 0), abi.BLK_TD_WHEEL: ("tdw", 0), abi.BLK_PLANE_R: ("plr", 0),
            abi.BLK_PLANE_Z: ("plz", 0)}[bid]


class Problem:
    """Stacked residuals / Jacobian of one window through `api.eval_factors` (api: the CPU oracle in the tests)."""

    def __init__(self, api, snap):
        self.api, self.snap0 = api, snap
        self.free, self.table, self.nd, self.sizes = _block_table(snap)
        L = len(snap["para_feature"])
        fconst = np.asarray(snap.get("feature_const", np.zeros(L)), bool)
        used = np.zeros(L, bool)
        used[np.asarray(snap["vis_feature_index"], int)] = True
        self.lm_free = used & ~fconst
        self.lm_col = np.full(L, -1)
        self.lm_col[self.lm_free] = self.nd + np.arange(self.lm_free.sum())
        self.n = self.nd + int(self.lm_free.sum())

    def _cols(self, b):
        o = self.table.get(b)
        return None if o is None else np.arange(o, o + self.sizes[b[0]][1])

    def _optional(self, snap):
        """Plane / anchor factors through the stand-alone evaluators (pinned on their own in tests/test_optional_oracle.py)."""
        out = []
        fc = int(snap.get("frame_count", abi.WINDOW_SIZE))
        if snap.get("plane") is not None and fc > 0:
            e = abi.plane_eval(self.api.lib, self.api.prefix, None, np.asarray(snap["pose"])[:fc], snap["ex_pose_wheel"],
                               snap.get("plane_R", [0, 0, 0, 1.0]), float(snap.get("plane_Z", 0.0)), snap["plane"]["noise_inv"])
            for i in range(fc):
                out.append((e["r"][i], e["J"][i], [(("pose", i), 6), (("exw", 0), 6), (("plr", 0), 3), (("plz", 0), 1)]))
        if snap.get("anchor") is not None:
            e = abi.anchor_eval(self.api.lib, self.api.prefix, None, np.asarray(snap["pose"])[:1], np.asarray(snap["anchor"]["pose"])[None],
                                float(snap["anchor"].get("sqrt_info", 120.0)))
            out.append((e["r"][0], e["J"][0], [(("pose", 0), 6)]))
        if gnss_factors_on(snap):       # through the stand-alone evaluator (pinned on its own in tests/test_gnss_oracle.py)
            gn, gs = snap["gnss"], snap["gnss_state"]
            e = abi.gnss_eval(self.api.lib, self.api.prefix, None, gn["obs"], gn.get("iono"), snap["pose"], snap["speed_bias"], gs["rcv_dt"], gs["rcv_ddt"],
                              gs["yaw_enu_local"], gs["anc_ecef"], gn["frame_dt"], gn["ddt_weight"])
            for k, o in enumerate(gn["obs"]):
                lo, fr = int(o["lower_idx"]), int(o["frame"])
                # the evaluator's 18 columns: P_i V_i P_j V_j (3 each: the leading columns of the pose / speed-bias blocks) dt ddt yaw anc
                Jb = np.zeros((2, 6 + 9 + 6 + 9 + 1 + 1 + 1 + 3))
                J = e["J"][k]
                Jb[:, 0:3], Jb[:, 6:9], Jb[:, 15:18], Jb[:, 21:24], Jb[:, 30:36] = J[:, 0:3], J[:, 3:6], J[:, 6:9], J[:, 9:12], J[:, 12:18]
                out.append((e["r"][k], Jb, [(("pose", lo), 6), (("sb", lo), 9), (("pose", lo + 1), 6), (("sb", lo + 1), 9), (("dt", 4 * fr + int(o["sys_idx"])), 1),
                                            (("ddt", fr), 1), (("yaw", 0), 1), (("anc", 0), 3)]))
            fdt = np.asarray(gn["frame_dt"], float)
            for k in range(4):
                for i in range(abi.WINDOW_SIZE):
                    out.append((e["r_dt_ddt"][k, i:i + 1], np.array([[-50.0, 50.0, -25.0 * fdt[i], -25.0 * fdt[i]]]),
                                [(("dt", 4 * i + k), 1), (("dt", 4 * (i + 1) + k), 1), (("ddt", i), 1), (("ddt", i + 1), 1)]))
            wgt = float(gn["ddt_weight"])
            for i in range(abi.WINDOW_SIZE):
                out.append((e["r_smooth"][i:i + 1], np.array([[wgt, -wgt]]), [(("ddt", i), 1), (("ddt", i + 1), 1)]))
        return out

    def evaluate(self, snap, want_jacobian=True):
        ev = self.api.eval_factors(snap, robustify=True)
        opt = self._optional(snap)
        cost = ev["cost"]      # (the whole objective: eval_factors counts the optional factors' cost too)
        if not want_jacobian:
            return cost, None, None
        rows_r, rows_J = [], []

        def add(r, Jb, blocks):
            Jrow = np.zeros((len(r), self.n))
            c0 = 0
            for b, w in blocks:
                cols = b if isinstance(b, np.ndarray) else self._cols(b)
                if cols is not None and len(cols):
                    Jrow[:, cols] = Jb[:, c0:c0 + len(cols)]
                c0 += w
            rows_r.append(r)
            rows_J.append(Jrow)

        s = snap
        for k in range(len(s["vis_imu_i"])):
            l = int(s["vis_feature_index"][k])
            lc = np.array([self.lm_col[l]]) if self.lm_col[l] >= 0 else np.zeros(0, int)
            add(ev["vis_r"][k], ev["vis_J"][k], [(("pose", int(s["vis_imu_i"][k])), 6), (("pose", int(s["vis_imu_j"][k])), 6), (("exc", 0), 6), (lc, 1), (("td", 0), 1)])
        for k, f in enumerate(s.get("imu_frame", [])):
            f = int(f)
            add(ev["imu_r"][k], ev["imu_J"][k], [(("pose", f), 6), (("sb", f), 9), (("pose", f + 1), 6), (("sb", f + 1), 9)])
        for k, f in enumerate(s.get("wheel_frame", [])):
            f = int(f)
            add(ev["wheel_r"][k], ev["wheel_J"][k], [(("pose", f), 6), (("pose", f + 1), 6), (("exw", 0), 6), (("sx", 0), 1), (("sy", 0), 1), (("sw", 0), 1), (("tdw", 0), 1)])
        pr = s.get("prior")
        if pr is not None:
            J0 = np.asarray(pr["J0"]).reshape(int(pr["n"]), int(pr["n"]))
            blocks = []
            order = np.argsort(pr["block_idx"])
            for q in order:
                b = _block_of_id(int(pr["block_id"][q]))
                blocks.append((b, self.sizes[b[0]][1]))
            # J0's columns are laid out by block_idx: reorder to the sorted-by-offset sequence used above
            # (a 4-wide prior block of the plane quaternion carries 4 columns; the solve's tangent Jacobian is the first three —
            #  OrientationSubsetParameterization::ComputeJacobian = [I3; 0] — and the residual uses all four differences)
            blocks = [(b, w) if b[0] != "plr" else (b, 4) for b, w in blocks]
            # (a constant block — the yaw, a constant extrinsic — keeps its columns out of J but its difference in the residual)
            perm = np.concatenate([np.arange(int(pr["block_idx"][q]), int(pr["block_idx"][q]) + (4 if _block_of_id(int(pr["block_id"][q]))[0] == "plr" else self.sizes[_block_of_id(int(pr["block_id"][q]))[0]][1])) for q in order])
            add(ev["prior_r"], J0[:, perm], blocks)
        for r_, J_, blocks in opt:
            add(r_, J_, blocks)
        return cost, np.concatenate(rows_r), np.vstack(rows_J)

    # ---- manifold
    def plus(self, snap, delta):
        s = dict(snap)
        s["pose"] = np.array(snap["pose"], float).copy()
        s["speed_bias"] = np.array(snap["speed_bias"], float).copy()
        s["ex_pose"] = np.array(snap["ex_pose"], float).copy()
        s["ex_pose_wheel"] = np.array(snap["ex_pose_wheel"], float).copy()
        s["ix_wheel"] = np.array(snap["ix_wheel"], float).copy()
        s["para_feature"] = np.array(snap["para_feature"], float).copy()
        if snap.get("gnss_state") is not None:
            s["gnss_state"] = {k: np.array(v, float).copy() for k, v in snap["gnss_state"].items()}

        def pose_plus(x, d, mask=None):
            d = np.array(d, float)
            if mask is not None:
                d = np.where(np.asarray(mask, bool), 0.0, d)
            dq = np.array([0.5 * d[3], 0.5 * d[4], 0.5 * d[5], 1.0])      # Utility::deltaQ: [theta / 2, 1], normalised after the product
            q = synth.qmul(x[3:], dq)
            return np.concatenate([x[:3] + d[:3], q / np.linalg.norm(q)])

        for b in self.free:
            d = delta[self._cols(b)]
            if b[0] == "pose":
                s["pose"][b[1]] = pose_plus(s["pose"][b[1]], d)
            elif b[0] == "sb":
                s["speed_bias"][b[1]] = s["speed_bias"][b[1]] + d
            elif b[0] == "exc":
                s["ex_pose"] = pose_plus(s["ex_pose"], d, snap.get("ex_cam_mask"))
            elif b[0] == "exw":
                s["ex_pose_wheel"] = pose_plus(s["ex_pose_wheel"], d, snap.get("ex_wheel_mask"))
            elif b[0] == "td":
                s["td"] = float(snap["td"]) + d[0]
            elif b[0] == "tdw":
                s["td_wheel"] = float(snap["td_wheel"]) + d[0]
            elif b[0] == "plr":      # OrientationSubsetParameterization({2}): component 2 held in Plus
                dq = np.array([0.5 * d[0], 0.5 * d[1], 0.0, 1.0])
                q = synth.qmul(np.asarray(snap.get("plane_R", [0, 0, 0, 1.0]), float), dq)
                s["plane_R"] = q / np.linalg.norm(q)
            elif b[0] == "plz":
                s["plane_Z"] = float(snap.get("plane_Z", 0.0)) + d[0]
            elif b[0] == "anc":
                s["gnss_state"]["anc_ecef"] += d
            elif b[0] == "yaw":
                s["gnss_state"]["yaw_enu_local"] += d[0]
            elif b[0] == "dt":
                s["gnss_state"]["rcv_dt"].reshape(-1)[b[1]] += d[0]
            elif b[0] == "ddt":
                s["gnss_state"]["rcv_ddt"][b[1]] += d[0]
            else:
                s["ix_wheel"][{"sx": 0, "sy": 1, "sw": 2}[b[0]]] += d[0]
        s["para_feature"][self.lm_free] += delta[self.nd:]
        return s

    def ambient(self, snap):
        """The reduced program's parameter vector (free blocks, global coordinates)."""
        parts = []
        for b in self.free:
            if b[0] == "pose":
                parts.append(np.asarray(snap["pose"])[b[1]])
            elif b[0] == "sb":
                parts.append(np.asarray(snap["speed_bias"])[b[1]])
            elif b[0] == "exc":
                parts.append(np.asarray(snap["ex_pose"]))
            elif b[0] == "exw":
                parts.append(np.asarray(snap["ex_pose_wheel"]))
            elif b[0] == "td":
                parts.append([float(snap["td"])])
            elif b[0] == "tdw":
                parts.append([float(snap["td_wheel"])])
            elif b[0] == "plr":
                parts.append(np.asarray(snap.get("plane_R", [0, 0, 0, 1.0]), float))
            elif b[0] == "plz":
                parts.append([float(snap.get("plane_Z", 0.0))])
            elif b[0] == "anc":
                parts.append(np.asarray(snap["gnss_state"]["anc_ecef"], float))
            elif b[0] == "yaw":
                parts.append([float(snap["gnss_state"]["yaw_enu_local"])])
            elif b[0] == "dt":
                parts.append([np.asarray(snap["gnss_state"]["rcv_dt"], float).reshape(-1)[b[1]]])
            elif b[0] == "ddt":
                parts.append([np.asarray(snap["gnss_state"]["rcv_ddt"], float)[b[1]]])
            else:
                parts.append([np.asarray(snap["ix_wheel"])[{"sx": 0, "sy": 1, "sw": 2}[b[0]]]])
        parts.append(np.asarray(snap["para_feature"])[self.lm_free])
        return np.concatenate([np.ravel(p) for p in parts])


def solve(api, snap, max_num_iterations=8, initial_radius=1e4, function_tolerance=1e-6, gradient_tolerance=1e-10,
          parameter_tolerance=1e-8, min_relative_decrease=1e-3, fail_chol_iter=0):
    """Returns dict(accepted, cost_history, radius_history, mu_history, termination, snap). fail_chol_iter: the fault injection
    of tests/test_gpu_branches.py (first factorisation of that iteration fails)."""
    P = Problem(api, snap)
    x = snap
    cost, r, J = P.evaluate(x)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(axis=0)))        # jacobi scaling, fixed at iteration 0
    radius, mu = initial_radius, MIN_MU
    reuse = False
    x_norm = np.linalg.norm(P.ambient(x))
    accepted, costs, radii, mus = [0], [cost], [radius], [mu]
    invalid, it, termination = 0, 0, 0
    Js = D = g = gt = gn_t = None
    alpha = 0.0
    while True:
        if it >= max_num_iterations:
            termination = 0
            break
        if np.abs(J.T @ r).max() <= gradient_tolerance:
            termination = 3
            break
        if radius < MIN_RADIUS:
            termination = 4
            break
        it += 1
        if not reuse:
            Js = J * scale
            D = np.sqrt(np.clip((Js * Js).sum(axis=0), MIN_DIAG, MAX_DIAG))
            g = Js.T @ r
            gt = g / D
            Jg = Js @ (gt / D)
            alpha = (gt @ gt) / (Jg @ Jg)
            A = Js.T @ Js
            solved, inject = False, fail_chol_iter == it
            while mu < MAX_MU:
                try:
                    if inject:
                        inject = False
                        raise np.linalg.LinAlgError("injected")
                    Lc = np.linalg.cholesky(A + mu * np.diag(D * D))
                    y = -np.linalg.solve(Lc.T, np.linalg.solve(Lc, g))
                    if np.isfinite(y).all():
                        solved = True
                        break
                except np.linalg.LinAlgError:
                    pass
                mu *= MU_INC
            if not solved:
                termination = 4
                break
            gn_t = D * y
            reuse = True
        g_norm, gn_norm = np.linalg.norm(gt), np.linalg.norm(gn_t)
        if gn_norm <= radius:
            step_t = gn_t
        elif alpha * g_norm >= radius:
            step_t = -(radius / g_norm) * gt
        else:
            a = -alpha * gt                       # Cauchy point; b = Gauss-Newton point
            b_dot_a = gn_t @ a
            a2, bma2 = a @ a, (gn_t - a) @ (gn_t - a)
            c = b_dot_a - a2
            d = np.sqrt(c * c + bma2 * (radius * radius - a2))
            beta = (d - c) / bma2 if c <= 0 else (radius * radius - a2) / (d + c)
            step_t = a + beta * (gn_t - a)
        step_norm = np.linalg.norm(step_t)
        step = step_t / D
        Jstep = Js @ step
        model_change = -(Jstep @ (r + 0.5 * Jstep))
        if not model_change > 0.0:
            accepted.append(0); costs.append(cost); radii.append(radius); mus.append(mu)
            invalid += 1
            if invalid >= 5:
                termination = 4
                break
            mu *= MU_INC
            reuse = False
            continue
        invalid = 0
        xc = P.plus(x, scale * step)
        cand, _, _ = P.evaluate(xc, want_jacobian=False)
        if not np.isfinite(cand):
            cand = np.finfo(float).max
        step_amb = np.linalg.norm(P.ambient(xc) - P.ambient(x))
        if step_amb <= parameter_tolerance * (x_norm + parameter_tolerance):
            accepted.append(0); costs.append(cost); radii.append(radius); mus.append(mu)
            termination = 2
            break
        dcost = cost - cand
        if abs(dcost) <= function_tolerance * cost:
            accepted.append(0); costs.append(cost); radii.append(radius); mus.append(mu)
            termination = 1
            break
        rho = dcost / model_change
        if rho > min_relative_decrease:
            x, cost = xc, cand
            x_norm = np.linalg.norm(P.ambient(x))
            _, r, J = P.evaluate(x)
            if rho < 0.25:
                radius *= 0.5
            if rho > 0.75:
                radius = max(radius, 3.0 * step_norm)
            mu = max(MIN_MU, 2.0 * mu / MU_INC)
            reuse = False
            accepted.append(1)
        else:
            radius *= 0.5
            reuse = True
            accepted.append(0)
        costs.append(cost); radii.append(radius); mus.append(mu)
    return dict(accepted=accepted, cost_history=costs, radius_history=radii, mu_history=mus, termination=termination,
                iterations=it, final_cost=cost, final_radius=radius, snap=x)
