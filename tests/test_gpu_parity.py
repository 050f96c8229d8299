"""GPU parity tests proper: the HIP back end through the C ABI vs the CPU oracle on the same seeded
inputs. FP64 everywhere; tolerances are written next to each assertion (north_star: "results match
the reference solve on the same window to a stated float tolerance; bit-exact index bookkeeping")."""
import os

import numpy as np
import pytest

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def window_with_prior(oracle, seed, L, use_wheel=True):
    scn = synth.Scenario(seed=seed, n_landmarks=L, use_wheel=use_wheel)
    resA = oracle.solve(scn.window(0), abi.MARGIN_OLD)
    stB = synth.shift_state_for_next_window(scn, resA["state"], 1)
    return scn, scn.window(1, state=stB, prior=resA["prior"])


@pytest.mark.parametrize("robust", [False, True])
def test_factor_blocks_match_oracle(be, oracle, robust):
    _, snap = window_with_prior(oracle, 51, 300)
    snap["ix_wheel"] = np.array([1.01, 0.98, 1.02])
    snap["td"], snap["td_wheel"] = 0.003, -0.004
    want = oracle.eval_factors(snap, robustify=robust)
    got = be.eval_factors(snap, robustify=robust)
    # 1e-12 relative on every visual / wheel block (same FP64 formulas, different association order)
    for k in ("vis_r", "vis_J", "wheel_r", "wheel_J"):
        assert relerr(got[k], want[k]) < 1e-12, k
    # IMU: sqrt_info from a covariance of condition ~1e12 -> 1e-9
    for k in ("imu_r", "imu_J"):
        assert relerr(got[k], want[k]) < 1e-9, k
    assert relerr(got["prior_r"], want["prior_r"]) < 1e-11
    if robust:    # the oracle's cost is always the robustified objective 1/2 sum rho(|r|^2)
        assert abs(got["cost"] - want["cost"]) < 1e-10 * want["cost"]


def check_solve(be, oracle, snap, flag, loose=1.0):
    """Stated FP64 tolerances of a whole optimization() call vs the oracle (measured deviations on
    MI355X are 2-4 orders of magnitude below them, see tools/diag_parity.py):
      accept/reject sequence, iteration count, termination reason : identical
      cost after every iteration : 1e-6 relative (the transient iterations drop the cost by 1e4)
      final cost                 : 1e-9 relative
      ATE of the 11 window poses : 1e-8 m ; rotations 1e-9 rad ; speed/bias 1e-7 ; inverse depths 1e-7 rel
    """
    want = oracle.solve(snap, flag)
    got = be.solve(snap, flag)
    sw, sg = want["summary"], got["summary"]
    assert sg["iterations"] == sw["iterations"]
    assert sg["accepted"] == sw["accepted"]
    assert sg["termination"] == sw["termination"]
    np.testing.assert_allclose(sg["cost_history"], sw["cost_history"], rtol=loose * 1e-6)
    # (loose: a multiplier for windows that stop before they have settled, stated by the caller)
    assert abs(sg["final_cost"] - sw["final_cost"]) < loose * 1e-9 * sw["final_cost"]
    ate = np.sqrt(((got["state"]["pose"][:, :3] - want["state"]["pose"][:, :3]) ** 2).sum(axis=1).mean())
    assert ate < loose * 1e-8, ate
    for i in range(abi.NFRAMES):
        dq = synth.qmul(synth.qinv(want["state"]["pose"][i, 3:]), got["state"]["pose"][i, 3:])
        assert 2 * np.linalg.norm(dq[:3]) < loose * 1e-9
    assert np.abs(got["state"]["speed_bias"] - want["state"]["speed_bias"]).max() < loose * 1e-7
    for k in ("ex_pose", "ex_pose_wheel", "ix_wheel"):
        assert np.abs(got["state"][k] - want["state"][k]).max() < loose * 1e-7, k
    np.testing.assert_allclose(got["feature"], want["feature"], rtol=loose * 1e-7, atol=1e-12)
    return want, got


def test_solve_cfg1_no_prior_no_wheel(be, oracle):
    """BASELINE.json configs[0]: 10-kf VIO window, 200 landmarks, no wheel, no prior."""
    scn = synth.Scenario(seed=20250708, n_landmarks=200, use_wheel=False)
    check_solve(be, oracle, scn.window(0), abi.MARGIN_NONE)


def test_solve_cfg2_wheel_prior_2k(be, oracle):
    """BASELINE.json configs[1]: 10-kf VI-wheel window, 2k landmarks, with a marginalisation prior."""
    _, snap = window_with_prior(oracle, 20250709, 2000)
    want, got = check_solve(be, oracle, snap, abi.MARGIN_OLD)
    pw, pg = want["prior"], got["prior"]
    assert pg["block_id"].tolist() == pw["block_id"].tolist()
    assert pg["block_size"].tolist() == pw["block_size"].tolist()
    assert pg["block_idx"].tolist() == pw["block_idx"].tolist()
    np.testing.assert_allclose(pg["x0"], pw["x0"], rtol=0, atol=1e-6)
    # sqrt factors are unique only up to an orthogonal transform: compare J0^T J0 and J0^T r0
    Aw, Ag = pw["J0"].T @ pw["J0"], pg["J0"].T @ pg["J0"]
    bw, bg = pw["J0"].T @ pw["r0"], pg["J0"].T @ pg["r0"]
    assert np.abs(Ag - Aw).max() < 1e-7 * np.abs(Aw).max()
    assert np.abs(bg - bw).max() < 1e-6 * max(np.abs(bw).max(), 1.0)


def test_solve_cfg3_10k_landmarks(be, oracle):
    """BASELINE.json configs[2] on one GPU: 10-kf VI-wheel window, 10 000 landmarks (K ~ 47k factors,
    157 landmark tiles), with a marginalisation prior."""
    _, snap = window_with_prior(oracle, 20250710, 10000)
    check_solve(be, oracle, snap, abi.MARGIN_OLD)


def test_prior_square_root_modes(oracle):
    """marg_sqrt = 0 (eigen-decomposition, the reference's construction) and 1 (pivoted LDL^T, default)
    give the same information J0^T J0, J0^T r0 — 1e-9 relative to max|A'| — and both are valid square
    roots of the oracle's thresholded A'."""
    _, snap = window_with_prior(oracle, 20250711, 600)
    want = oracle.solve(snap, abi.MARGIN_OLD)["prior"]
    Aw, bw = want["J0"].T @ want["J0"], want["J0"].T @ want["r0"]
    got = {}
    for mode in (0, 1):
        o = abi.default_options()
        o.marg_sqrt = mode
        pg = gf.Backend(device=0, options=o).solve(snap, abi.MARGIN_OLD)["prior"]
        assert pg["block_id"].tolist() == want["block_id"].tolist() and pg["n"] == want["n"]
        A, b = pg["J0"].T @ pg["J0"], pg["J0"].T @ pg["r0"]
        assert np.abs(A - Aw).max() < 1e-9 * np.abs(Aw).max(), mode
        assert np.abs(b - bw).max() < 1e-7 * max(1.0, np.abs(bw).max()), mode
        got[mode] = (A, b)
    assert np.abs(got[0][0] - got[1][0]).max() < 1e-9 * np.abs(Aw).max()
    # using either prior in the next window gives the same solve: windows 1 and 2 of a run (its own scenario: thirteen keyframes), window 2
    # from the same shifted state with the eigen prior, the LDL^T prior and the oracle's — identical accept / reject sequences, poses
    # within 1e-8 m, final cost 1e-9
    scn = synth.Scenario(seed=20250713, n_landmarks=600, use_wheel=True, n_kf=13)
    r0 = oracle.solve(scn.window(0), abi.MARGIN_OLD)
    w1 = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    r1w = oracle.solve(w1, abi.MARGIN_OLD)
    st2 = synth.shift_state_for_next_window(scn, r1w["state"], 2)
    res = {"oracle": oracle.solve(scn.window(2, state=st2, prior=r1w["prior"]), abi.MARGIN_OLD)}
    for mode in (0, 1):
        o = abi.default_options()
        o.marg_sqrt = mode
        bes = gf.Backend(device=0, options=o)
        r1 = bes.solve(w1, abi.MARGIN_OLD)
        res[mode] = bes.solve(scn.window(2, state=st2, prior=r1["prior"]), abi.MARGIN_OLD)
        bes.close()
    for key in (1, "oracle"):
        a, b = res[0], res[key]
        assert a["summary"]["accepted"] == b["summary"]["accepted"] and a["summary"]["iterations"] == b["summary"]["iterations"], key
        # (5e-8 m: the window is the second of a chain that started from two different solvers' priors; what differs is the drift of the
        #  positions along the trajectory — 2e-8 m at its far end against the oracle's chain, 1e-9 between the two square roots)
        assert np.abs(a["state"]["pose"][:, :3] - b["state"]["pose"][:, :3]).max() < (5e-8 if key == "oracle" else 1e-8), key
        # (two square roots of one information matrix agree in J0^T J0 and J0^T r0, not in |r0|^2 — the prior's CONSTANT term, which the
        #  smallest kept eigenvalues of a 1e14-conditioned A' amplify: LDL^T against eigen 6e-7 of the cost here, the device's eigen
        #  square root against the oracle's 3e-4. The constant moves no pose: what the solve does with any of the three priors is the
        #  same decrease from the same start)
        da, db = a["summary"]["initial_cost"] - a["summary"]["final_cost"], b["summary"]["initial_cost"] - b["summary"]["final_cost"]
        assert abs(da - db) < 1e-9 * max(a["summary"]["initial_cost"], 1.0), key


def test_second_new_and_passthrough(be, oracle):
    _, snap = window_with_prior(oracle, 53, 400)
    want, got = check_solve(be, oracle, snap, abi.MARGIN_SECOND_NEW)
    pw, pg = want["prior"], got["prior"]
    assert pg["n"] == pw["n"] == snap["prior"]["n"] - 6
    assert pg["block_id"].tolist() == pw["block_id"].tolist()
    Aw, Ag = pw["J0"].T @ pw["J0"], pg["J0"].T @ pg["J0"]
    assert np.abs(Ag - Aw).max() < 1e-7 * np.abs(Aw).max()


def test_constant_landmarks_and_frozen_window(be, oracle):
    """estimate_flag==1 landmarks are constant (estimator.cpp:3352); stationary mode freezes every
    pose / speed-bias block (estimator.cpp:3294-3307)."""
    scn = synth.Scenario(seed=54, n_landmarks=300, use_wheel=True)
    snap = scn.window(0)
    snap["feature_const"] = (np.arange(300) % 3 == 0).astype(np.uint8)
    want, got = check_solve(be, oracle, snap, abi.MARGIN_OLD)
    np.testing.assert_array_equal(got["feature"][::3], snap["para_feature"][::3])
    snap2 = scn.window(0)
    snap2["pose_const"] = np.ones(11, np.uint8)
    snap2["sb_const"] = np.ones(11, np.uint8)
    want, got = check_solve(be, oracle, snap2, abi.MARGIN_NONE)
    np.testing.assert_allclose(got["state"]["pose"][:, :3], snap2["pose"][:, :3], atol=1e-12)


def test_batch_equals_single_and_is_deterministic(be, oracle):
    snaps = [synth.Scenario(seed=60 + k, n_landmarks=150 + 40 * k, use_wheel=bool(k % 2)).window(0) for k in range(5)]
    single = [be.solve(s, abi.MARGIN_OLD) for s in snaps]
    batch = be.solve_batch(snaps, abi.MARGIN_OLD)
    again = be.solve_batch(snaps, abi.MARGIN_OLD)
    for a, b, c in zip(single, batch, again):
        np.testing.assert_array_equal(a["state"]["pose"], b["state"]["pose"])      # fixed-order reductions
        np.testing.assert_array_equal(b["state"]["pose"], c["state"]["pose"])
        np.testing.assert_array_equal(a["feature"], b["feature"])
        assert a["summary"]["cost_history"] == b["summary"]["cost_history"]


@pytest.mark.parametrize("B", [40, 131])
def test_large_batch_throughput_path(be, oracle, B):
    """Batches of >= 32 windows take the throughput path (k_dense_raw with one lane per window, the dense factors on
    a second stream beside the visual kernels, one k_visblock workgroup per window, three start-frame groups instead of
    eleven in the landmark elimination), batches of >= 128 are additionally solved as two halves side by side on two pairs
    of streams. Every window comes out the same whatever its position and neighbours in the batch, bit for bit and repeatably
    (fixed-order reductions); against the single-window path — whose kernels are shaped for latency and add the same
    terms in another grouping — identical accept / reject sequences and the tolerances of a settled solve. Windows with and
    without prior / wheel / LiDAR block."""
    snaps = [synth.Scenario(seed=160 + k, n_landmarks=120 + 30 * k, use_wheel=bool(k % 2)).window(0) for k in range(4)]
    scn = synth.Scenario(seed=166, n_landmarks=400, use_wheel=True)
    r0 = be.solve(scn.window(0), abi.MARGIN_OLD)
    with_prior = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    snaps += [with_prior, dict(with_prior, lio=synth.lidar_block(scn, 1, n=700, seed=4, outliers=0.05))]
    n = len(snaps)
    single = [be.solve(s, abi.MARGIN_OLD) for s in snaps]
    big = [snaps[i % n] for i in range(B)]
    batch = be.solve_batch(big, abi.MARGIN_OLD)
    again = be.solve_batch(big, abi.MARGIN_OLD)
    for i, (b, c) in enumerate(zip(batch, again)):
        first = batch[i % n]                        # the same window at another place of the batch (other half, other neighbours)
        for other in (first, c):
            assert b["summary"] == other["summary"]
            np.testing.assert_array_equal(b["state"]["pose"], other["state"]["pose"])
            np.testing.assert_array_equal(b["state"]["speed_bias"], other["state"]["speed_bias"])
            np.testing.assert_array_equal(b["feature"], other["feature"])
            np.testing.assert_array_equal(b["prior"]["J0"], other["prior"]["J0"])
            np.testing.assert_array_equal(b["prior"]["r0"], other["prior"]["r0"])
    for i in range(n):
        a, b = single[i], batch[i]
        assert a["summary"]["accepted"] == b["summary"]["accepted"] and a["summary"]["termination"] == b["summary"]["termination"]
        np.testing.assert_allclose(b["summary"]["cost_history"], a["summary"]["cost_history"], rtol=1e-7)
        assert abs(a["summary"]["final_cost"] - b["summary"]["final_cost"]) < 1e-9 * a["summary"]["final_cost"]
        assert np.abs(a["state"]["pose"] - b["state"]["pose"]).max() < 1e-8
        assert np.abs(a["state"]["speed_bias"] - b["state"]["speed_bias"]).max() < 1e-7
        np.testing.assert_allclose(b["feature"], a["feature"], rtol=1e-7, atol=1e-12)
        Aa, Ab = a["prior"]["J0"].T @ a["prior"]["J0"], b["prior"]["J0"].T @ b["prior"]["J0"]
        assert np.abs(Aa - Ab).max() < 1e-7 * np.abs(Aa).max()


def test_graph_replay_is_bit_identical(oracle):
    """gfbe_options.use_graph = 1: the third solve of a resident batch replays the captured hipGraph — same bits as the
    eager launches."""
    snaps = [synth.Scenario(seed=170 + k, n_landmarks=200, use_wheel=True).window(0) for k in range(3)]
    outs = []
    for use_graph in (0, 1):
        o = abi.default_options()
        o.use_graph = use_graph
        b = gf.Backend(device=0, options=o)
        batch = b.batch_upload(snaps)
        for _ in range(4):
            batch.solve(abi.MARGIN_OLD)
        outs.append(batch.download())
        batch.free()
        b.close()
    for a, g in zip(*outs):
        np.testing.assert_array_equal(a["state"]["pose"], g["state"]["pose"])
        np.testing.assert_array_equal(a["feature"], g["feature"])
        np.testing.assert_array_equal(a["prior"]["J0"], g["prior"]["J0"])
        assert a["summary"]["cost_history"] == g["summary"]["cost_history"]


def test_bad_inputs_fail_loudly(be, oracle):
    """gfbe_status instead of garbage (SURVEY §8b "Errors" row): a factor with imu_j <= imu_i or an out-of-range landmark
    index is GFBE_BAD_INPUT (3) before anything is launched; a NaN state makes every linear solve fail ->
    GFBE_NUMERICAL_FAILURE (2), on the device and in the oracle alike."""
    snap = synth.Scenario(seed=9, n_landmarks=100, use_wheel=True).window(0)
    bad = dict(snap, vis_imu_j=snap["vis_imu_j"].copy())
    bad["vis_imu_j"][3] = bad["vis_imu_i"][3]
    with pytest.raises(RuntimeError, match="status 3"):
        be.solve(bad, abi.MARGIN_OLD)
    bad = dict(snap, vis_feature_index=snap["vis_feature_index"].copy())
    bad["vis_feature_index"][0] = len(snap["para_feature"]) + 5
    with pytest.raises(RuntimeError, match="status 3"):
        be.solve(bad, abi.MARGIN_OLD)
    nan = dict(snap, pose=snap["pose"].copy())
    nan["pose"][4, 1] = np.nan
    with pytest.raises(RuntimeError, match="status 2"):
        be.solve(nan, abi.MARGIN_OLD)
    with pytest.raises(RuntimeError, match="status 2"):
        oracle.solve(nan, abi.MARGIN_OLD)
    # the context stays usable after a failure
    check_solve(be, oracle, snap, abi.MARGIN_OLD)


def test_partial_window_and_empty_visual(be, oracle):
    """frame_count < WINDOW_SIZE (estimator.cpp:3391: no marginalisation) and a window without
    any visual factor (IMU + wheel only)."""
    scn = synth.Scenario(seed=70, n_landmarks=100, use_wheel=True)
    snap = scn.window(0)
    keep = snap["vis_imu_j"] <= 6
    for k in list(snap):
        if k.startswith("vis_"):
            snap[k] = snap[k][keep]
    snap["frame_count"] = 6
    snap["imu"], snap["imu_frame"] = snap["imu"][:6], snap["imu_frame"][:6]
    snap["wheel"], snap["wheel_frame"] = snap["wheel"][:6], snap["wheel_frame"][:6]
    check_solve(be, oracle, snap, abi.MARGIN_OLD)
    snap2 = scn.window(0)
    for k in list(snap2):
        if k.startswith("vis_"):
            snap2[k] = snap2[k][:0]
    check_solve(be, oracle, snap2, abi.MARGIN_NONE)


def test_preintegration_matches_oracle(be, oracle):
    scn = synth.Scenario(seed=80, n_landmarks=5, use_wheel=True)
    noise = [synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W]
    got = be.preintegrate_imu(scn.imu_raw, scn.ba_est, scn.bg_est, noise)
    want = oracle.preintegrate_imu(scn.imu_raw, scn.ba_est, scn.bg_est, noise)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-16)
    gotw = be.preintegrate_wheel(scn.wheel_raw, [1.0, 1.0, 1.0, 0.0], [synth.VEL_N_WHEEL, synth.GYR_N_WHEEL])
    wantw = oracle.preintegrate_wheel(scn.wheel_raw, [1.0, 1.0, 1.0, 0.0], [synth.VEL_N_WHEEL, synth.GYR_N_WHEEL])
    np.testing.assert_allclose(gotw, wantw, rtol=1e-10, atol=1e-16)
    # ragged: intervals of different lengths, including a single-sample one
    ragged = [(scn.imu_raw[0][0][:1], scn.imu_raw[0][1]), (scn.imu_raw[1][0][:7], scn.imu_raw[1][1]), scn.imu_raw[2]]
    np.testing.assert_allclose(be.preintegrate_imu(ragged, scn.ba_est, scn.bg_est, noise),
                               oracle.preintegrate_imu(ragged, scn.ba_est, scn.bg_est, noise), rtol=1e-10, atol=1e-16)


def test_noise_free_ate(be):
    """Size-independent property at the full bench size: a noise-free 2k-landmark window converges
    to ground truth (ATE -> 0) with the wheel extrinsic held (its vertical lever arm is unobservable)."""
    scn = synth.Scenario(seed=90, n_landmarks=2000, use_wheel=True, noise=False)
    truth = scn.truth_state(0)
    rng = np.random.default_rng(1)
    st = scn.truth_state(0)
    for i in range(1, abi.NFRAMES):
        st["pose"][i, :3] += rng.normal(0, 0.02, 3)
        q = synth.qmul(st["pose"][i, 3:], synth.so3_exp(rng.normal(0, np.deg2rad(0.5), 3)))
        st["pose"][i, 3:] = q / np.linalg.norm(q)
        st["speed_bias"][i, :3] += rng.normal(0, 0.05, 3)
    snap = scn.window(0, state=st)
    snap["ex_wheel_const"] = 1
    res = be.solve(snap, abi.MARGIN_OLD)
    ate = np.sqrt(((res["state"]["pose"][:, :3] - truth["pose"][:, :3]) ** 2).sum(axis=1).mean())
    assert ate < 5e-4, ate
    assert res["summary"]["final_cost"] < 1e-6 * res["summary"]["initial_cost"]


def test_invariants_over_a_chain_of_windows_at_bench_size(be):
    """Size-independent properties at the bench configuration (2k landmarks, wheel, prior), WITHOUT the oracle: five
    consecutive optimization() calls, each fed with the previous call's state and prior —
      * the cost never rises on an accepted step and stays where it was on a rejected one,
      * yaw and position of frame 0 are what they were before the call (double2vector's gauge fix),
      * J0^T J0 of the prior that comes back is symmetric positive semi-definite and its blocks are the reference's
        (poses 1..10 renamed 0..9 first),
      * a second solve from the solution makes no progress beyond 1e-3 of the cost."""
    scn = synth.Scenario(seed=123, n_landmarks=2000, use_wheel=True, n_kf=16)
    st, prior = None, None
    for k in range(5):
        snap = scn.window(k, state=st, prior=prior)
        res = be.solve(snap, abi.MARGIN_OLD)
        sm = res["summary"]
        hist = sm["cost_history"][: sm["iterations"] + 1]
        assert len(hist) >= 3
        for i in range(1, len(hist)):
            if sm["accepted"][i]:
                assert hist[i] < hist[i - 1]
            else:
                assert hist[i] == hist[i - 1]
        np.testing.assert_allclose(res["state"]["pose"][0, :3], snap["pose"][0, :3], atol=1e-12)
        Ra, Rb = synth.qrot(snap["pose"][0, 3:]), synth.qrot(res["state"]["pose"][0, 3:])
        assert abs(np.arctan2(Ra[1, 0], Ra[0, 0]) - np.arctan2(Rb[1, 0], Rb[0, 0])) < 1e-12
        pr = res["prior"]
        A = pr["J0"].T @ pr["J0"]
        assert np.abs(A - A.T).max() <= 1e-12 * np.abs(A).max() and np.linalg.eigvalsh(A).min() > -1e-9 * np.abs(A).max()
        assert pr["block_id"][:10].tolist() == list(range(10)) and pr["block_size"][:10].tolist() == [7] * 10
        again = be.solve(dict(snap, para_feature=res["feature"], **res["state"]), abi.MARGIN_NONE)
        # (the state that comes back is re-anchored: the prior is not invariant under that rigid yaw / position shift of the
        #  window, so the second solve starts up to 3e-4 relative above the first one's final cost — measured)
        assert abs(again["summary"]["initial_cost"] - sm["final_cost"]) < 2e-3 * sm["final_cost"]
        assert again["summary"]["final_cost"] <= again["summary"]["initial_cost"]
        assert again["summary"]["initial_cost"] - again["summary"]["final_cost"] < 1e-2 * sm["final_cost"]
        st = synth.shift_state_for_next_window(scn, res["state"], k + 1)
        prior = pr


def test_factor_blocks_bit_identical_to_host_build(be):
    """k_prep forms the square-root information matrices with one wave per factor (64 lanes sharing the LU inverse and the
    Cholesky of the inverse); every entry goes through the operations of the one-thread sqrt_info_from_cov in the same order,
    so the whitened IMU / wheel residuals and Jacobians of the device equal, bit for bit, those of the same header compiled for
    the host (tests/host_shim.cpp; no fused multiply-add in either build)."""
    import test_device_math_host as tdm
    shim = tdm.build_shim()          # (skips without hipcc)
    for seed in (41, 42):
        scn = synth.Scenario(seed=seed, n_landmarks=120, use_wheel=True)
        snap = scn.window(0)
        for rob in (False, True):
            got, want = be.eval_factors(snap, robustify=rob), tdm.shim_eval(shim, snap, rob)
            # ... and so do the visual blocks (the fused evaluation + Huber corrector of the linearisation kernels): what the CPU
            # suite pins against the oracle (tests/test_device_math_host.py) IS what the GPU computes
            for k in ("imu_r", "imu_J", "wheel_r", "wheel_J", "vis_r", "vis_J"):
                assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k


def test_split_batch_sorts_its_windows_by_size_and_returns_them_in_place(oracle):
    """A batch that is solved as several parts side by side holds its windows sorted by their number of visual factors (every
    part's grids then fit its own windows, gfbe_host.cpp::upload_halves); results, landmark counts and priors come back at the
    caller's places: 160 windows of five sizes in a shuffled order, four parts, against the same batch solved as ONE part — the
    same kernels, so bit for bit."""
    rng = np.random.default_rng(5)
    kinds = [synth.Scenario(seed=170 + k, n_landmarks=60 + 70 * k, use_wheel=bool(k % 2)).window(0) for k in range(5)]
    pick = rng.integers(0, 5, 160)
    snaps = [kinds[int(i)] for i in pick]
    o1, o4 = abi.default_options(), abi.default_options()
    o1.split_batch, o4.split_batch = 0, 4
    one, four = gf.Backend(device=0, options=o1), gf.Backend(device=0, options=o4)
    want = one.solve_batch(snaps, abi.MARGIN_OLD)
    batch = four.batch_upload(snaps)
    batch.solve(abi.MARGIN_OLD)
    got = batch.download()
    for i in range(len(snaps)):
        assert four.lib.gfbe_batch_feature_count(batch.h, i) == len(snaps[i]["para_feature"])
    batch.free()
    for i, (a, b) in enumerate(zip(want, got)):
        assert len(b["feature"]) == len(snaps[i]["para_feature"])
        assert a["summary"] == b["summary"]
        np.testing.assert_array_equal(a["state"]["pose"], b["state"]["pose"])
        np.testing.assert_array_equal(a["state"]["speed_bias"], b["state"]["speed_bias"])
        np.testing.assert_array_equal(a["feature"], b["feature"])
        np.testing.assert_array_equal(a["prior"]["J0"], b["prior"]["J0"])
        assert a["prior"]["block_id"].tolist() == b["prior"]["block_id"].tolist()
    one.close(); four.close()


def test_constant_td_that_differs_from_the_observations_stamps(be, oracle):
    """td is held constant (the shipped configuration) but is not the td the observations were stamped with: every observation is
    shifted by (td - td_obs) x velocity (projectionTwoFrameOneCamFactor.cpp:60-61). Host-fed batches do that on the HOST while they
    pack (two doubles per factor cross PCIe, velocities only for the landmarks of frame 0: upload_one); the marginalisation's td /
    extrinsic columns still see the velocities. Against the oracle, alone and inside a throughput batch."""
    _, snap = window_with_prior(oracle, 53, 300)
    snap["td"] = 0.004
    snap["vis_td_j"] = np.asarray(snap["vis_td_j"], float) + 0.001 * (np.arange(len(snap["vis_td_j"])) % 3)      # (stamps that differ between observations)
    assert snap.get("td_const", 1) and np.abs(snap["td"] - snap["vis_td_j"]).min() > 0
    want, got = check_solve(be, oracle, snap, abi.MARGIN_OLD)
    assert relerr(got["prior"]["J0"].T @ got["prior"]["J0"], want["prior"]["J0"].T @ want["prior"]["J0"]) < 1e-7
    ids = got["prior"]["block_id"].tolist()
    assert abi.BLK_TD in ids and abi.BLK_EX_CAM in ids                  # (the prior keeps td and the extrinsic: their columns need the velocities)
    batch = be.solve_batch([snap] * 33, abi.MARGIN_OLD)
    assert batch[7]["summary"]["accepted"] == got["summary"]["accepted"]
    np.testing.assert_allclose(batch[7]["summary"]["cost_history"], got["summary"]["cost_history"], rtol=1e-7)
    assert np.abs(batch[7]["state"]["pose"] - got["state"]["pose"]).max() < 1e-8
    assert relerr(batch[7]["prior"]["J0"].T @ batch[7]["prior"]["J0"], got["prior"]["J0"].T @ got["prior"]["J0"]) < 1e-7


def test_assembled_system_of_the_two_kernel_sets_entry_by_entry(oracle):
    """The normal equations of the FIRST linearisation (one iteration: both runs linearise at the same state) — H (lower triangle), g,
    E after the landmark elimination — from the small-batch kernel set (k_lin_small / k_dense, k_schur_visblock_small, k_assemble)
    and from the throughput set (k_vis<0> + k_dense_tp + k_prior_tp, k_schur, k_visasm): the same quantities summed in other
    groupings, entry by entry through gfbe_debug_vector. Wheel and prior present; the window sits at place 5 of a 33-window batch."""
    o = abi.default_options()
    o.max_num_iterations = 1
    be1 = gf.Backend(device=0, options=o)
    _, snap = window_with_prior(oracle, 57, 260)
    rows = list(range(0, abi.DENSE_DIM, 7)) + [66, 67, 72, 73, 100, 164, 186]
    got = []
    for B, w in ((1, 0), (33, 5)):
        b = be1.batch_upload([snap] * B)
        b.solve(abi.MARGIN_NONE)
        H = np.array([b.debug_vector(1000 + r, w) for r in rows])
        E = np.array([b.debug_vector(2000 + r, w)[:73] for r in range(0, 74, 3)])
        g = b.debug_vector(3, w)
        got.append((H, E, g, b.download()[w]["summary"]))
        b.free()
    be1.close()
    (H1, E1, g1, s1), (H2, E2, g2, s2) = got
    assert s1["iterations"] == s2["iterations"] == 1 and s1["accepted"] == s2["accepted"]
    tri = np.array([[c <= r for c in range(abi.DENSE_DIM)] for r in rows])      # (H holds its lower triangle)
    assert np.abs(H1[tri]).max() > 1.0 and np.abs(g1).max() > 1.0
    assert np.abs(H1[tri] - H2[tri]).max() < 1e-11 * np.abs(H1[tri]).max()
    assert np.abs(E1 - E2).max() < 1e-11 * np.abs(E1).max()
    assert np.abs(g1 - g2).max() < 1e-11 * np.abs(g1).max()
