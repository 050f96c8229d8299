"""Invariants that pin the oracle's solver / re-anchoring / marginalisation restatement
(SURVEY.md §8c item iii): there is no Ceres here to compare against (parity UNPINNED)."""
import numpy as np
import pytest

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth

# oracle tangent layout (oracle/gfo_solver.cpp)
T_POSE = lambda k: 6 * k          # noqa: E731
T_EX, T_TD = 66, 72
T_SB = lambda k: 73 + 9 * k       # noqa: E731
T_EXW, T_SX, T_SY, T_SW, T_TDW = 172, 178, 179, 180, 181


def numpy_normal_equations(snap, ev):
    """H = sum J'J, g = sum J'r from the block-CSR factor outputs (robustified)."""
    H, g = np.zeros((abi.DENSE_DIM, abi.DENSE_DIM)), np.zeros(abi.DENSE_DIM)
    L = len(snap["para_feature"])
    Hll, gl, Hpl = np.zeros(L), np.zeros(L), np.zeros((L, 73))
    for k in range(len(snap["vis_imu_i"])):
        i, j, l = snap["vis_imu_i"][k], snap["vis_imu_j"][k], snap["vis_feature_index"][k]
        cols = np.r_[T_POSE(i) + np.arange(6), T_POSE(j) + np.arange(6), T_EX + np.arange(6), T_TD]
        J = ev["vis_J"][k][:, np.r_[0:18, 19]]
        w = ev["vis_J"][k][:, 18]
        r = ev["vis_r"][k]
        H[np.ix_(cols, cols)] += J.T @ J
        g[cols] += J.T @ r
        Hll[l] += w @ w
        gl[l] += w @ r
        Hpl[l, cols] += J.T @ w
    for k, i in enumerate(snap["imu_frame"]):
        cols = np.r_[T_POSE(i) + np.arange(6), T_SB(i) + np.arange(9), T_POSE(i + 1) + np.arange(6), T_SB(i + 1) + np.arange(9)]
        J, r = ev["imu_J"][k], ev["imu_r"][k]
        H[np.ix_(cols, cols)] += J.T @ J
        g[cols] += J.T @ r
    for k, i in enumerate(snap.get("wheel_frame", [])):
        cols = np.r_[T_POSE(i) + np.arange(6), T_POSE(i + 1) + np.arange(6), T_EXW + np.arange(6), T_SX, T_SY, T_SW, T_TDW]
        J, r = ev["wheel_J"][k], ev["wheel_r"][k]
        H[np.ix_(cols, cols)] += J.T @ J
        g[cols] += J.T @ r
    return H, g, Hll, gl, Hpl


def free_all(snap):
    s = dict(snap)
    s.update(ex_cam_const=0, ex_wheel_const=0, ix_wheel_const=0, td_const=0, td_wheel_const=0)
    return s


def test_linearize_matches_blockwise_sum(oracle):
    scn = synth.Scenario(seed=31, n_landmarks=50, use_wheel=True)
    snap = free_all(scn.window(0))
    ev = oracle.eval_factors(snap, robustify=True)
    H, g, Hll, gl, Hpl = numpy_normal_equations(snap, ev)
    lin = oracle.linearize(snap)
    scale = np.abs(H).max()
    assert np.abs(lin["H"] - H).max() < 1e-12 * scale
    assert np.abs(lin["g"] - g).max() < 1e-12 * np.abs(g).max()
    np.testing.assert_allclose(lin["Hll"], Hll, rtol=1e-12)
    np.testing.assert_allclose(lin["gl"], gl, rtol=1e-10, atol=1e-9)
    assert np.abs(lin["Hpl"] - Hpl).max() < 1e-12 * np.abs(Hpl).max()
    assert abs(lin["cost"] - ev["cost"]) < 1e-12 * ev["cost"]


def test_constant_blocks_are_removed(oracle):
    scn = synth.Scenario(seed=32, n_landmarks=30, use_wheel=True)
    snap = scn.window(0)           # m3dgr.yaml flags: camera extrinsic, intrinsics, td, td_wheel constant
    lin = oracle.linearize(snap)
    for a in list(range(T_EX, T_EX + 6)) + [T_TD, T_SX, T_SY, T_SW, T_TDW]:
        assert np.all(lin["H"][a] == 0) and np.all(lin["H"][:, a] == 0) and lin["g"][a] == 0
    assert np.all(lin["Hpl"][:, T_EX:T_EX + 7] == 0)
    assert np.any(lin["H"][T_EXW:T_EXW + 6] != 0)


def test_noise_free_window_converges_to_truth(oracle):
    scn = synth.Scenario(seed=33, n_landmarks=120, use_wheel=True, noise=False)
    truth = scn.truth_state(0)
    rng = np.random.default_rng(0)
    st = scn.truth_state(0)
    st["speed_bias"][:, 3:6], st["speed_bias"][:, 6:9] = scn.ba_est, scn.bg_est
    for i in range(1, abi.NFRAMES):          # frame 0 stays at truth => gauge-fixed comparison
        st["pose"][i, :3] += rng.normal(0, 0.02, 3)
        q = synth.qmul(st["pose"][i, 3:], synth.so3_exp(rng.normal(0, np.deg2rad(0.5), 3)))
        st["pose"][i, 3:] = q / np.linalg.norm(q)
        st["speed_bias"][i, :3] += rng.normal(0, 0.05, 3)
    snap = scn.window(0, state=st)
    # the component of the wheel lever arm along the (vertical) rotation axis is unobservable for
    # planar motion; hold the extrinsic constant for a gauge-free comparison with ground truth
    snap["ex_wheel_const"] = 1
    snap["para_feature"] = snap["para_feature"] * (1 + rng.normal(0, 0.1, len(snap["para_feature"])))
    o = oracle.with_options(max_num_iterations=15)
    res = o.solve(snap, abi.MARGIN_NONE)
    sm = res["summary"]
    assert sm["final_cost"] < 1e-3 and sm["final_cost"] < 1e-9 * sm["initial_cost"]
    ate = np.sqrt(((res["state"]["pose"][:, :3] - truth["pose"][:, :3]) ** 2).sum(axis=1).mean())
    assert ate < 2e-4, ate
    for i in range(abi.NFRAMES):
        dq = synth.qmul(synth.qinv(truth["pose"][i, 3:]), res["state"]["pose"][i, 3:])
        assert 2 * np.linalg.norm(dq[:3]) < 2e-4


def test_cost_monotone_and_gauge_fixed(oracle):
    scn = synth.Scenario(seed=34, n_landmarks=150, use_wheel=True)
    snap = scn.window(0)
    res = oracle.solve(snap, abi.MARGIN_NONE)
    sm = res["summary"]
    hist, acc = sm["cost_history"], sm["accepted"]
    for k in range(1, len(hist)):
        if acc[k]:
            assert hist[k] < hist[k - 1]
        else:
            assert hist[k] == hist[k - 1]
    assert sm["final_cost"] == hist[-1]
    # double2vector: position and yaw of frame 0 are pinned to their pre-solve values (estimator.cpp:2515-2546)
    np.testing.assert_allclose(res["state"]["pose"][0, :3], snap["pose"][0, :3], atol=1e-12)
    y0 = np.arctan2(*synth.qrot(snap["pose"][0, 3:])[[1, 0], 0])
    y1 = np.arctan2(*synth.qrot(res["state"]["pose"][0, 3:])[[1, 0], 0])
    assert abs(y0 - y1) < 1e-12
    for i in range(abi.NFRAMES):
        assert abs(np.linalg.norm(res["state"]["pose"][i, 3:]) - 1) < 1e-12


def test_reanchor_is_a_rigid_yaw_translation(oracle):
    scn = synth.Scenario(seed=35, n_landmarks=10)
    a = scn.initial_state(0)
    b = scn.truth_state(0)
    out = oracle.reanchor(a, b)
    # relative poses are preserved
    for i in range(1, abi.NFRAMES):
        Rb0, Rbi = synth.qrot(b["pose"][0, 3:]), synth.qrot(b["pose"][i, 3:])
        Ro0, Roi = synth.qrot(out["pose"][0, 3:]), synth.qrot(out["pose"][i, 3:])
        assert np.abs(Rb0.T @ Rbi - Ro0.T @ Roi).max() < 1e-12
        d_b = Rb0.T @ (b["pose"][i, :3] - b["pose"][0, :3])
        d_o = Ro0.T @ (out["pose"][i, :3] - out["pose"][0, :3])
        assert np.abs(d_b - d_o).max() < 1e-12
    np.testing.assert_allclose(out["pose"][0, :3], a["pose"][0, :3], atol=1e-14)


def numpy_marginalize_old(snap, ev, prior_ev=None):
    """Dense Schur complement of {pose0, sb0, landmarks starting at frame 0} computed with numpy
    from the factor blocks (marginalization_factor.cpp:183-292), in the oracle's canonical order."""
    sel = np.where(snap["vis_imu_i"] == 0)[0]
    lms = sorted(set(snap["vis_feature_index"][sel].tolist()))
    keep_ids = []
    touched = set([0, 11, 1, 12, 22, 27, 23, 24, 25, 26, 28])
    for k in sel:
        touched.add(int(snap["vis_imu_j"][k]))
    keep_ids = [b for b in sorted(touched) if b not in (0, 11)]
    idx = {0: 0, 11: 6}
    pos = 15
    for l in lms:
        idx[("l", l)] = pos
        pos += 1
    m = pos
    for b in keep_ids:
        idx[b] = pos
        pos += abi.block_local_size(b)
    A, bb = np.zeros((pos, pos)), np.zeros(pos)

    def add(J, r, cols):
        A[np.ix_(cols, cols)] += J.T @ J
        bb[cols] += J.T @ r
    J, r = ev["imu_J"][0], ev["imu_r"][0]
    add(J, r, np.r_[idx[0] + np.arange(6), idx[11] + np.arange(9), idx[1] + np.arange(6), idx[12] + np.arange(9)])
    J, r = ev["wheel_J"][0], ev["wheel_r"][0]
    add(J, r, np.r_[idx[0] + np.arange(6), idx[1] + np.arange(6), idx[23] + np.arange(6), idx[24], idx[25], idx[26], idx[28]])
    for k in sel:
        j, l = int(snap["vis_imu_j"][k]), int(snap["vis_feature_index"][k])
        cols = np.r_[idx[0] + np.arange(6), idx[j] + np.arange(6), idx[22] + np.arange(6), idx[("l", l)], idx[27]]
        add(ev["vis_J"][k], ev["vis_r"][k], cols)
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    w, V = np.linalg.eigh(Amm)
    winv = np.where(w > 1e-8, 1.0 / w, 0.0)
    Ainv = (V * winv) @ V.T
    Ap = A[m:, m:] - A[m:, :m] @ Ainv @ A[:m, m:]
    bp = bb[m:] - A[m:, :m] @ Ainv @ bb[:m]
    return keep_ids, Ap, bp, np.abs(A).max(), np.abs(bb).max()


def test_marginalize_old_against_numpy_schur(oracle):
    scn = synth.Scenario(seed=36, n_landmarks=80, use_wheel=True)
    snap = scn.window(0)
    ev = oracle.eval_factors(snap, robustify=True)
    keep_ids, Ap, bp, a_scale, b_scale = numpy_marginalize_old(snap, ev)
    prior, A, b, rc = oracle.marginalize(snap, abi.MARGIN_OLD)
    assert rc == 0 and prior["valid"] == 1
    # block ids are shifted: pose i -> i-1, sb i -> i-1
    want_ids = [(k - 1 if k < 22 else k) for k in keep_ids]
    assert prior["block_id"].tolist() == want_ids
    assert prior["n"] == Ap.shape[0] == 86
    # A' = Arr - Arm Amm^-1 Amr cancels numbers of size a_scale (IMU information ~1e10): roundoff
    # of the two independent evaluations is relative to that scale
    sc = a_scale
    assert np.abs(A - Ap).max() < 1e-11 * a_scale, (np.abs(A - Ap).max(), a_scale)
    assert np.abs(b - bp).max() < 1e-11 * max(b_scale, a_scale * 1e-2)
    # sqrt factorisation identities (commented asserts at marginalization_factor.cpp:306-307),
    # exact on the eigen-space above eps
    J0, r0 = prior["J0"], prior["r0"]
    w, V = np.linalg.eigh(0.5 * (Ap + Ap.T))
    keep = w > 1e-8
    Aplus = (V[:, keep] * w[keep]) @ V[:, keep].T
    bplus = V[:, keep] @ (V[:, keep].T @ bp)
    assert np.abs(J0.T @ J0 - Aplus).max() < 1e-11 * sc
    assert np.abs(J0.T @ r0 - bplus).max() < 1e-11 * max(b_scale, a_scale * 1e-2)
    # x0 = parameter values at marginalisation time, old slots
    off = 0
    for bid_new, size in zip(prior["block_id"], prior["block_size"]):
        bid_old = bid_new + 1 if bid_new < 22 else bid_new
        if bid_old < 11:
            want = snap["pose"][bid_old]
        elif bid_old < 22:
            want = snap["speed_bias"][bid_old - 11]
        elif bid_old == 22:
            want = snap["ex_pose"]
        elif bid_old == 23:
            want = snap["ex_pose_wheel"]
        elif bid_old in (24, 25, 26):
            want = snap["ix_wheel"][bid_old - 24:bid_old - 23]
        else:
            want = [snap["td"] if bid_old == 27 else snap["td_wheel"]]
        np.testing.assert_array_equal(prior["x0"][off:off + size], np.asarray(want, float))
        off += size


def numpy_pivoted_ldlt_sqrt(Ap, bp, eps=1e-8):
    """Independent statement of the product's default square root: diagonally pivoted LDL^T with pivots > eps,
    written as successive Schur complements on index sets (no in-place elimination loops like the C++ ones)."""
    n = len(bp)
    M, rhs = 0.5 * (Ap + Ap.T), bp.copy()
    left = list(range(n))
    J0, r0 = np.zeros((n, n)), np.zeros(n)
    for k in range(n):
        dg = np.array([M[i, i] for i in left])
        p = left[int(np.argmax(dg))]
        if not M[p, p] > eps:
            break
        d = M[p, p]
        col = np.zeros(n)
        col[left] = M[left, p] / d
        J0[k] = np.sqrt(d) * col
        r0[k] = rhs[p] / np.sqrt(d)
        left.remove(p)
        M[np.ix_(left, left)] -= d * np.outer(col[left], col[left])
        rhs[left] -= col[left] * (r0[k] * np.sqrt(d))
    return J0, r0


def test_ldlt_square_root_mode_of_the_oracle(oracle):
    """gfo_options.marg_sqrt = 1 (the product's default, NOT the reference's construction; used by bench.py's like-for-like CPU leg):
    the same information as the eigen form, and row for row the numpy statement above."""
    scn = synth.Scenario(seed=36, n_landmarks=80, use_wheel=True)
    snap = scn.window(0)
    pe, A, b, rc = oracle.marginalize(snap, abi.MARGIN_OLD)
    pl, A1, b1, rc1 = oracle.with_options(marg_sqrt=1).marginalize(snap, abi.MARGIN_OLD)
    assert rc == 0 and rc1 == 0
    assert pl["block_id"].tolist() == pe["block_id"].tolist() and np.array_equal(pl["x0"], pe["x0"]) and pl["block_idx"].tolist() == pe["block_idx"].tolist()
    sc = np.abs(A).max()
    # marg_sqrt = 1 is "the product's algorithm": the frame-0 landmarks (diagonal block) eliminated first, then the 15 dense dims —
    # the same Schur complement as the reference's whole-Amm pseudo-inverse (every eigenvalue of Amm exceeds eps here)
    # (A' cancels numbers of the un-reduced information's size — the inertial factor's ~1e10 —: roundoff is relative to that, as above)
    _, _, _, a_scale, b_scale = numpy_marginalize_old(snap, oracle.eval_factors(snap, robustify=True))
    assert np.abs(A1 - A).max() < 1e-11 * a_scale and np.abs(b1 - b).max() < 1e-11 * max(b_scale, 1e-2 * a_scale)
    A, b = A1, b1
    assert np.abs(pl["J0"].T @ pl["J0"] - pe["J0"].T @ pe["J0"]).max() < 1e-11 * a_scale
    assert np.abs(pl["J0"].T @ pl["r0"] - pe["J0"].T @ pe["r0"]).max() < 1e-11 * max(b_scale, 1e-2 * a_scale)
    J0n, r0n = numpy_pivoted_ldlt_sqrt(A, b)
    rank = int((np.abs(J0n).sum(axis=1) > 0).sum())
    assert rank == int((np.abs(pl["J0"]).sum(axis=1) > 0).sum()) and 70 <= rank <= 86
    # the trailing pivots are Schur complements that have cancelled eight or more digits of A': their rows agree to what is left
    np.testing.assert_allclose(pl["J0"][:60], J0n[:60], rtol=1e-7, atol=1e-9 * np.sqrt(sc))
    np.testing.assert_allclose(pl["r0"][:60], r0n[:60], rtol=1e-6, atol=1e-7 * np.abs(r0n).max())
    # and a whole solve that carries the LDL^T prior forward ends where the eigen one does
    r0_ = oracle.solve(snap, abi.MARGIN_OLD)
    r1_ = oracle.with_options(marg_sqrt=1).solve(snap, abi.MARGIN_OLD)
    nxt = [scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"]) for r in (r0_, r1_)]
    c0, c1 = oracle.solve(nxt[0], abi.MARGIN_NONE)["summary"], oracle.solve(nxt[1], abi.MARGIN_NONE)["summary"]
    assert c0["accepted"] == c1["accepted"] and abs(c0["final_cost"] - c1["final_cost"]) < 1e-7 * c0["final_cost"]


def test_prior_chain_old_then_second_new(oracle):
    scn = synth.Scenario(seed=37, n_landmarks=300, use_wheel=True)   # enough tracks from frame 0 to reach frame 10
    resA = oracle.solve(scn.window(0), abi.MARGIN_OLD)
    prior = resA["prior"]
    assert prior is not None and prior["n"] == 86
    stB = synth.shift_state_for_next_window(scn, resA["state"], 1)
    snapB = scn.window(1, state=stB, prior=prior)
    evB = oracle.eval_factors(snapB)
    # kept blocks that did not move since marginalisation reproduce r0 exactly
    moved = np.abs(evB["prior_r"] - prior["r0"]).max()
    assert moved < 1e-6 * max(1.0, np.abs(prior["r0"]).max())
    resB = oracle.solve(snapB, abi.MARGIN_SECOND_NEW)
    assert resB["summary"]["final_cost"] < resB["summary"]["initial_cost"]
    p2 = resB["prior"]
    assert p2 is not None and p2["n"] == prior["n"] - 6
    # pose 9 (second newest) is dropped; the prior never holds pose 10, so nothing moves to slot 9
    assert p2["block_id"].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 22, 23, 24, 25, 26, 27, 28]
    # MARGIN_SECOND_NEW on a prior that does not touch pose 9 leaves nothing to do
    snapC = dict(snapB)
    snapC["prior"] = None
    prC, _, _, rc = oracle.marginalize(snapC, abi.MARGIN_SECOND_NEW)
    assert prC is None and rc == 1


def test_feature_bookkeeping_bit_exact(oracle):
    scn = synth.Scenario(seed=38, n_landmarks=60)
    fl = scn.feature_list(0, extra_short=25)
    fl["estimate_flag"][::7] = 1
    for only0 in (False, True):
        want = synth.build_visual_factors_np(fl)
        if only0:
            m = want["vis_imu_i"] == 0
            for k in list(want):
                if k.startswith("vis_"):
                    want[k] = want[k][m]
        got = oracle.build_visual_factors(fl, only0)
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    lam = want["para_feature"].copy()
    lam[3] = -0.5
    est, flag = oracle.set_depth(fl, lam)
    long = np.where(fl["n_obs"] >= 4)[0]
    assert flag[long[3]] == 2 and (flag[long[:3]] == 1).all() and (flag[fl["n_obs"] < 4] == 0).all()
    np.testing.assert_array_equal(est[long], 1.0 / lam)


def test_second_new_with_an_invalid_prior_that_lists_the_second_newest_pose(oracle):
    """estimator.cpp:3600, 3622-3632: last_marginalization_info exists, is NOT valid (a marginalisation that had nothing to drop
    keeps its block list, marginalization_factor.cpp:204-210, 310-330) and lists Pose[WINDOW_SIZE - 1]: the else-branch marginalises
    a PoseAnchorFactor on Pose[0] with drop set {Pose[0]} — nothing is kept, the new info is valid and EMPTY. Without
    Pose[WINDOW_SIZE - 1] in the list the old (invalid) info simply stays."""
    scn = synth.Scenario(seed=31, n_landmarks=120, use_wheel=True)
    r0 = oracle.solve(scn.window(0), abi.MARGIN_OLD)
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    bad = dict(r0["prior"], valid=0)
    assert abi.BLK_POSE0 + abi.WINDOW_SIZE - 1 in bad["block_id"].tolist()
    wh = abi.WindowHolder(dict(snap, prior=bad))
    pr = abi.PriorHolder()
    pr.c.valid, pr.c.n, pr.c.n_blocks = 7, 7, 7          # (whatever the caller left there)
    st, feat, sm = abi.State(), np.zeros(wh.n_feature), abi.Summary()
    import ctypes as C
    rc = oracle.lib.gfo_solve_window(oracle.head, C.byref(wh.c), abi.MARGIN_SECOND_NEW, C.byref(st), abi._pd(feat), C.byref(pr.c), C.byref(sm))
    assert rc in (abi.OK, abi.NO_CONVERGENCE)
    assert (pr.c.valid, pr.c.n, pr.c.n_blocks) == (1, 0, 0)
    # the solve itself ran without a prior factor (estimator.cpp:3004: only a valid info is added)
    plain = oracle.solve(dict(snap, prior=None), abi.MARGIN_NONE)
    assert abi.summary_to_dict(sm)["cost_history"] == plain["summary"]["cost_history"]
