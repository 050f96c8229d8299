"""Branches of the hot path the standard windows never take ("GPU-untested branches"), each through the
C ABI on the GPU against the CPU oracle with the tolerances of tests/test_gpu_parity.py::check_solve:
  * every optional block free (camera extrinsic, td, wheel intrinsics, td_wheel: estimator.cpp:3024-3161) with non-zero
    PoseSubsetParameterization masks (pose_subset_parameterization.cpp:27-64: masked in Plus only);
  * double2vector()'s gimbal-lock branch (estimator.cpp:2523-2532): frame-0 pitch within 1 degree of 90;
  * the mu-retry of DoglegStrategy after a failed linear solve, with the in-kernel rebuild of E (fault injection:
    gfbe_options.test_fail_chol_iter makes the first factorisation of one iteration "fail" in both implementations);
  * USE_IMU = 0 (estimator.cpp:3018-3022): no IMU factors, speed-bias blocks absent, Pose[0] constant.
"""
import os

import numpy as np
import pytest

from _gfbe_import import gf
from test_gpu_parity import check_solve, window_with_prior

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


def all_free(snap, masks=True):
    s = dict(snap)
    s.update(ex_cam_const=0, ex_wheel_const=0, ix_wheel_const=0, td_const=0, td_wheel_const=0)
    if masks:   # CameraExtrinsicAdjustType / WheelExtrinsicAdjustType style masks (1 = component held in Plus)
        s["ex_cam_mask"] = np.array([0, 0, 1, 0, 0, 0], np.uint8)
        s["ex_wheel_mask"] = np.array([0, 0, 1, 1, 1, 0], np.uint8)
    s["ix_wheel"] = np.array([1.01, 0.99, 1.02])
    s["td"], s["td_wheel"] = 0.002, -0.003
    return s


@pytest.mark.parametrize("flag", [abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW])
def test_all_blocks_free_with_subset_masks(be, oracle, flag):
    _, snap = window_with_prior(oracle, 61, 400)
    snap = all_free(snap)
    # this window stops on rejected steps three accepted steps after a cost of 1e8 (not settled): the tolerances of a settled
    # solve x 100
    want, got = check_solve(be, oracle, snap, flag, loose=100.0)
    # the free blocks really moved, and the masked components did not (Plus zeroes them; the Jacobian still sees them)
    assert abs(got["state"]["td"] - snap["td"]) > 1e-9 and np.abs(got["state"]["ix_wheel"] - snap["ix_wheel"]).max() > 1e-9
    assert got["state"]["ex_pose"][2] == snap["ex_pose"][2]
    assert got["state"]["ex_pose_wheel"][2] == snap["ex_pose_wheel"][2]
    assert abs(got["state"]["td"] - want["state"]["td"]) < 1e-9 and abs(got["state"]["td_wheel"] - want["state"]["td_wheel"]) < 1e-9
    pw, pg = want["prior"], got["prior"]
    assert pg["block_id"].tolist() == pw["block_id"].tolist() and pg["block_idx"].tolist() == pw["block_idx"].tolist()
    Aw, Ag = pw["J0"].T @ pw["J0"], pg["J0"].T @ pg["J0"]
    assert np.abs(Ag - Aw).max() < 1e-6 * np.abs(Aw).max()


def test_all_blocks_free_first_window_no_prior(be, oracle):
    scn = synth.Scenario(seed=62, n_landmarks=300, use_wheel=True)
    check_solve(be, oracle, all_free(scn.window(0), masks=False), abi.MARGIN_OLD)


def test_reanchor_gimbal_lock_branch(be, oracle):
    """Frame 0 pitched to 89.5 degrees: |pitch| within 1 degree of 90 switches double2vector() from the yaw difference to
    rot_diff = R0 * R00^T (estimator.cpp:2523-2532). The window is otherwise the standard one (the factors then pull frame 0
    back: two iterations are enough to move it, and few enough for the two implementations not to drift apart)."""
    scn = synth.Scenario(seed=63, n_landmarks=300, use_wheel=True)
    snap = scn.window(0)
    R0 = synth.rz(0.3) @ synth.ry(np.deg2rad(89.5)) @ synth.rx(0.02)
    q = synth.rot2q(R0)
    snap["pose"] = snap["pose"].copy()
    snap["pose"][0, 3:] = q / np.linalg.norm(q)
    o2 = oracle.with_options(max_num_iterations=2)
    opt = abi.default_options()
    opt.max_num_iterations = 2
    be2 = gf.Backend(device=0, options=opt)
    want, got = o2.solve(snap, abi.MARGIN_NONE), be2.solve(snap, abi.MARGIN_NONE)
    assert got["summary"]["accepted"] == want["summary"]["accepted"] and got["summary"]["iterations"] == want["summary"]["iterations"]
    np.testing.assert_allclose(got["summary"]["cost_history"], want["summary"]["cost_history"], rtol=1e-9)
    # the gauge fix ran in its gimbal branch on both sides: frame 0 keeps its ORIGINAL full rotation, not just the yaw
    assert 2 * np.linalg.norm(synth.qmul(synth.qinv(snap["pose"][0, 3:]), got["state"]["pose"][0, 3:])[:3]) < 1e-9
    assert np.abs(got["state"]["pose"][:, :3] - want["state"]["pose"][:, :3]).max() < 1e-8
    for i in range(abi.NFRAMES):
        dq = synth.qmul(synth.qinv(want["state"]["pose"][i, 3:]), got["state"]["pose"][i, 3:])
        assert 2 * np.linalg.norm(dq[:3]) < 1e-9
    assert np.abs(got["state"]["speed_bias"] - want["state"]["speed_bias"]).max() < 1e-7
    be2.close()


def _failing(oracle, fail_iter):
    """A backend and an oracle whose options carry the test hook gfbe_options.test_fail_chol_iter."""
    opt = abi.default_options()
    opt.test_fail_chol_iter = fail_iter
    return gf.Backend(device=0, options=opt), oracle.with_options(test_fail_chol_iter=fail_iter)


@pytest.mark.parametrize("fail_iter", [1, 3])
def test_mu_retry_after_failed_linear_solve(be, oracle, fail_iter):
    """The first Cholesky of iteration `fail_iter` is declared failed: mu goes 1e-8 -> 1e-7, k_solve rebuilds E from the
    landmark rows for the new mu inside the kernel and factorises again; the later iterations carry the larger mu."""
    _, snap = window_with_prior(oracle, 64, 500)
    plain = oracle.solve(snap, abi.MARGIN_OLD)
    bef, orf = _failing(oracle, fail_iter)
    want, got = check_solve(bef, orf, snap, abi.MARGIN_OLD)
    assert want["summary"]["cost_history"] != plain["summary"]["cost_history"]      # the retry changed the iteration (larger damping)
    # a batch mixes windows: the retry of one must not leak into its neighbours' partials
    res = bef.solve_batch([snap, snap, snap], abi.MARGIN_OLD)
    for r in res:
        assert r["summary"]["cost_history"] == got["summary"]["cost_history"]
    bef.close()


def test_no_imu_pose0_constant(be, oracle):
    """USE_IMU = 0: no IMUFactor, no speed-bias blocks in the problem, Pose[0] fixed (estimator.cpp:3018-3022); the wheel
    odometry and the camera carry the window."""
    scn = synth.Scenario(seed=65, n_landmarks=300, use_wheel=True)
    snap = scn.window(0)
    snap["imu"] = np.zeros((0, abi.IMU_DOUBLES))
    snap["imu_frame"] = np.zeros(0, np.int32)
    pc = np.zeros(abi.NFRAMES, np.uint8)
    pc[0] = 1
    snap["pose_const"] = pc
    want, got = check_solve(be, oracle, snap, abi.MARGIN_OLD)
    assert np.array_equal(got["state"]["speed_bias"], snap["speed_bias"])            # untouched blocks come back as they went in
    assert got["prior"] is not None and abi.BLK_SB0 not in got["prior"]["block_id"].tolist()


def test_solver_time_cap_stops_on_the_device(be, oracle):
    """gfbe_options.max_solver_time_in_seconds (Solver::Options::max_solver_time_in_seconds = SOLVER_TIME, estimator.cpp:3369-3376):
    a cap far below one iteration stops the solve after the first step with termination 0 / NO_CONVERGENCE, no host sync
    involved; the default 0 never stops early."""
    scn = synth.Scenario(seed=66, n_landmarks=300, use_wheel=True)
    snap = scn.window(0)
    opt = abi.default_options()
    opt.max_solver_time_in_seconds = 1e-7
    bec = gf.Backend(device=0, options=opt)
    capped, full = bec.solve(snap, abi.MARGIN_NONE), be.solve(snap, abi.MARGIN_NONE)
    assert capped["summary"]["iterations"] < full["summary"]["iterations"]
    assert capped["summary"]["termination"] == 0 and capped["status"] == abi.NO_CONVERGENCE
    s = full["perf"]
    assert s["ms_solve"] > 0 and s["bytes_uploaded"] > 0 and s["bytes_downloaded"] > 0
    bec.close()


@pytest.mark.parametrize("name", ["A", "B", "cfg1", "free_masks", "retry2"])
def test_hip_backend_reproduces_independent_trust_region_loop(be, oracle, name):
    """The HIP path against tests/golden/dogleg_np.npz — the outputs of the independent numpy trust-region loop
    (tests/ceres_trust_region_np.py, from Ceres 1.14's published algorithm): same accept / reject sequence and termination,
    costs and final radius to the tolerances of tests/test_oracle_numpy.py::_check_against_loop."""
    from dogleg_cases import cases
    from test_oracle_numpy import _check_against_loop, _dogleg_fixture
    snap, kw = [(s, k) for n, s, k in cases(oracle) if n == name][0]
    bex = _failing(oracle, kw["fail_chol_iter"])[0] if kw.get("fail_chol_iter") else be
    _check_against_loop(bex.solve(snap, abi.MARGIN_NONE)["summary"], _dogleg_fixture(), name, final_rtol=1e-6 if name == "free_masks" else 1e-9)
    if bex is not be:
        bex.close()


def test_second_new_with_an_invalid_prior_that_lists_the_second_newest_pose(be, oracle):
    """estimator.cpp:3622-3632 (the PoseAnchorFactor else-branch of MARGIN_SECOND_NEW): an invalid last_marginalization_info that
    still lists Pose[WINDOW_SIZE - 1] is replaced by a valid, empty one; the solve runs without a prior factor. Same outcome as
    the oracle, alone and inside a batch next to a window with an ordinary prior."""
    import ctypes as C
    scn, snap = window_with_prior(oracle, 67, 300)
    bad = dict(snap["prior"], valid=0)
    odd = dict(snap, prior=bad)
    outs = []
    for lib, head, pre in ((be.lib, be.head, "gfbe_"), (oracle.lib, oracle.head, "gfo_")):
        wh = abi.WindowHolder(odd)
        pr = abi.PriorHolder()
        pr.c.valid, pr.c.n, pr.c.n_blocks = 7, 7, 7
        st, feat, sm = abi.State(), np.zeros(wh.n_feature), abi.Summary()
        f = getattr(lib, pre + "solve_window")
        f.restype = abi.c_i
        rc = f(head, C.byref(wh.c), abi.MARGIN_SECOND_NEW, C.byref(st), abi._pd(feat), C.byref(pr.c), C.byref(sm))
        assert rc in (abi.OK, abi.NO_CONVERGENCE)
        outs.append(((pr.c.valid, pr.c.n, pr.c.n_blocks), abi.summary_to_dict(sm)))
    assert outs[0][0] == outs[1][0] == (1, 0, 0)
    assert outs[0][1]["accepted"] == outs[1][1]["accepted"]
    assert abs(outs[0][1]["final_cost"] - outs[1][1]["final_cost"]) < 1e-9 * outs[1][1]["final_cost"]
    # in a batch: the ordinary window still gets its SECOND_NEW prior
    res = be.solve_batch([snap, odd], abi.MARGIN_SECOND_NEW)
    assert res[0]["prior"] is not None and res[0]["prior"]["n"] > 0
    assert res[1]["prior"] is not None and res[1]["prior"]["n"] == 0 and len(res[1]["prior"]["block_id"]) == 0


@pytest.mark.parametrize("reps,split", [(1, 1), (12, 1), (44, 2)])
def test_a_failing_window_is_its_own_failure_inside_a_batch(reps, split):
    """The windows of a batch are independent (include/gfbe.h, gfbe_solve_batch): a NaN state makes every linear solve of ITS window fail
    (GFBE_NUMERICAL_FAILURE in its summary, the worst status as the call's return value) and changes no bit of its neighbours' results
    — in the small-batch kernel set (3 windows), the throughput set (36) and a batch solved as two parts side by side (132: the
    failure sits in one part, the other is still unpacked)."""
    good = [synth.Scenario(seed=9 + k, n_landmarks=100 + 40 * k, use_wheel=True).window(0) for k in range(2)]
    nan = dict(good[0], pose=good[0]["pose"].copy())
    nan["pose"][4, 1] = np.nan
    o = abi.default_options()
    o.split_batch = split
    be = gf.Backend(device=0, options=o)
    try:
        alone = [be.solve(s, abi.MARGIN_OLD) for s in good]
        snaps = [good[0], nan, good[1]] * reps
        b = be.batch_upload(snaps)
        try:
            b.solve(abi.MARGIN_OLD)
            with pytest.raises(RuntimeError, match="status 2"):
                b.download()
            got = b.download(raise_on_failure=False)
        finally:
            b.free()
        for k, g in enumerate(got):
            if k % 3 == 1:
                assert g["status"] == abi.NUMERICAL_FAILURE
                continue
            a = alone[0 if k % 3 == 0 else 1]
            assert g["status"] in (abi.OK, abi.NO_CONVERGENCE) and g["summary"]["accepted"] == a["summary"]["accepted"]
            if reps == 1:      # (the small-batch kernels are the single window's own: the same bits)
                assert np.array_equal(g["state"]["pose"], a["state"]["pose"]) and np.array_equal(g["prior"]["J0"], a["prior"]["J0"])
            else:              # (the throughput kernels sum in another order)
                assert np.abs(g["state"]["pose"] - a["state"]["pose"]).max() < 1e-9
            assert np.array_equal(g["state"]["pose"], got[k % 3]["state"]["pose"])      # and the same bits wherever the window sits
    finally:
        be.close()


def test_failure_contract_of_solve_window(be):
    """include/gfbe.h, "FAILURE CONTRACT" (VERDICT round 5 item 9), through gfbe_solve_window itself: a FAILED CALL (status >= GFBE_BAD_INPUT)
    touches no output — state, inverse depths, the in-out prior and the summary keep the caller's bytes —, a numerically failed SOLVE
    (GFBE_NUMERICAL_FAILURE) has written all of them."""
    import ctypes as C
    snap = synth.Scenario(seed=9, n_landmarks=100, use_wheel=True).window(0)

    def call(s, flag):
        wh = abi.WindowHolder(s)
        st, pr, sm = abi.State(), abi.PriorHolder(), abi.Summary()
        feat = np.full(wh.n_feature, -7.25)
        C.memset(C.byref(st), 0x5A, C.sizeof(st))
        C.memset(C.byref(sm), 0x5A, C.sizeof(sm))
        pr.c.valid = 0
        pr.c.n = 12345
        f = be._fn("solve_window")
        f.restype = abi.c_i
        rc = f(be.head, C.byref(wh.c), int(flag), C.byref(st), abi._pd(feat), C.byref(pr.c), C.byref(sm))
        return rc, bytes(st), feat, pr, bytes(sm)

    rc, st, feat, pr, sm = call(snap, 7)                  # margin_flag out of range: the call fails
    assert rc >= abi.BAD_INPUT
    assert st == b"\x5a" * len(st) and sm == b"\x5a" * len(sm) and np.all(feat == -7.25) and pr.c.n == 12345 and pr.c.valid == 0
    nan = dict(snap, pose=snap["pose"].copy())
    nan["pose"][4, 1] = np.nan
    rc, st, feat, pr, sm = call(nan, abi.MARGIN_OLD)      # the solve runs and fails numerically: everything is written
    assert rc == abi.NUMERICAL_FAILURE
    assert st != b"\x5a" * len(st) and sm != b"\x5a" * len(sm) and not np.any(feat == -7.25) and pr.c.n != 12345
    rc, st, feat, pr, sm = call(snap, abi.MARGIN_OLD)     # and a good one
    assert rc in (abi.OK, abi.NO_CONVERGENCE) and pr.c.valid == 1 and not np.any(feat == -7.25)


def _first_factor_only(snap):
    """Every landmark keeps the first of its factors: tracks of two observations, the shortest a factor can be built from."""
    s = dict(snap)
    fi = np.asarray(snap["vis_feature_index"])
    _, first = np.unique(fi, return_index=True)
    keep = np.zeros(len(fi), bool)
    keep[first] = True
    for k in list(snap):
        if k.startswith("vis_"):
            s[k] = np.asarray(snap[k])[keep]
    return s


def test_shortest_tracks_one_factor_per_landmark(be, oracle):
    """Ragged down to the minimum: every landmark seen in exactly two frames (ONE factor: the landmark's share of the visual tile's
    observation loop is a single step, three of the four wave shares of its tile hold nothing). Against the oracle, alone and — the
    throughput kernels — inside a batch of 34."""
    _, snap = window_with_prior(oracle, 77, 300)
    short = _first_factor_only(snap)
    assert len(short["vis_feature_index"]) == len(np.unique(snap["vis_feature_index"]))
    want, got = check_solve(be, oracle, short, abi.MARGIN_OLD)
    many = be.solve_batch([short, snap] * 17, abi.MARGIN_OLD)
    for g in many[0::2]:
        assert g["summary"]["accepted"] == want["summary"]["accepted"]
        assert np.abs(g["state"]["pose"] - got["state"]["pose"]).max() < 1e-9
        np.testing.assert_allclose(g["feature"], got["feature"], rtol=1e-7, atol=1e-12)
