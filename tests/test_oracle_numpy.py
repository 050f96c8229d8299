"""Independent numpy re-derivation of the four residual models (SURVEY.md Appendix B equations,
rotation-matrix form) cross-checked against the C++ oracle at 1e-12 (SURVEY.md §8c item ii), and
of the two pre-integrators (synth.preintegrate_*_np, written from integration_base.h:63-137 /
wheel_integration_base.h:67-146) against oracle/gfo_preint.cpp."""
import numpy as np
import pytest

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
R = synth.qrot


def np_visual(snap, k):
    i, j, l = snap["vis_imu_i"][k], snap["vis_imu_j"][k], snap["vis_feature_index"][k]
    Pi, Ri = snap["pose"][i, :3], R(snap["pose"][i, 3:])
    Pj, Rj = snap["pose"][j, :3], R(snap["pose"][j, 3:])
    tic, ric = snap["ex_pose"][:3], R(snap["ex_pose"][3:])
    td, lam = snap["td"], snap["para_feature"][l]
    pi = snap["vis_pts_i"][k] - (td - snap["vis_td_i"][k]) * np.append(snap["vis_vel_i"][k], 0)
    pj = snap["vis_pts_j"][k] - (td - snap["vis_td_j"][k]) * np.append(snap["vis_vel_j"][k], 0)
    pc = ric.T @ (Rj.T @ (Ri @ (ric @ (pi / lam) + tic) + Pi - Pj) - tic)
    return 400.0 * (pc[:2] / pc[2] - pj[:2])


def np_imu_raw(snap, k, g=synth.G_NORM):
    rec = snap["imu"][k]
    dt, dp, dq, dv, lba, lbg = rec[0], rec[1:4], rec[4:8], rec[8:11], rec[11:14], rec[14:17]
    Jm = rec[17:17 + 225].reshape(15, 15)
    i = snap["imu_frame"][k]
    Pi, qi, Pj, qj = snap["pose"][i, :3], snap["pose"][i, 3:], snap["pose"][i + 1, :3], snap["pose"][i + 1, 3:]
    Vi, Bai, Bgi = np.split(snap["speed_bias"][i], 3)
    Vj, Baj, Bgj = np.split(snap["speed_bias"][i + 1], 3)
    dba, dbg = Bai - lba, Bgi - lbg
    th = Jm[3:6, 12:15] @ dbg
    dqc = np.append(th / 2, 1.0)
    cq = synth.qmul(dq, dqc / np.linalg.norm(dqc))
    cv = dv + Jm[6:9, 9:12] @ dba + Jm[6:9, 12:15] @ dbg
    cp = dp + Jm[0:3, 9:12] @ dba + Jm[0:3, 12:15] @ dbg
    G = np.array([0, 0, g])
    RiT = R(qi).T
    rp = RiT @ (0.5 * G * dt * dt + Pj - Pi - Vi * dt) - cp
    rq = 2 * synth.qmul(synth.qinv(cq), synth.qmul(synth.qinv(qi), qj))[:3]
    rv = RiT @ (G * dt + Vj - Vi) - cv
    cov = rec[17 + 225:].reshape(15, 15)
    return np.concatenate([rp, rq, rv, Baj - Bai, Bgj - Bgi]), cov


def np_wheel_raw(snap, k):
    rec = snap["wheel"][k]
    dp, dq = rec[1:4], rec[4:8]
    lsx, lsy, lsw, ltd = rec[8:12]
    lin_vel, lin_gyr, vel_1, gyr_1 = rec[12:15], rec[15:18], rec[18:21], rec[21:24]
    Jm = rec[24:42].reshape(6, 3)
    cov = rec[42:78].reshape(6, 6)
    i = snap["wheel_frame"][k]
    Pi, Ri, Pj, Rj = snap["pose"][i, :3], R(snap["pose"][i, 3:]), snap["pose"][i + 1, :3], R(snap["pose"][i + 1, 3:])
    tio, Rio = snap["ex_pose_wheel"][:3], R(snap["ex_pose_wheel"][3:])
    sx, sy, sw = snap["ix_wheel"]
    sv = np.diag([sx, sy, 1.0])
    cp = dp + Jm[0:3, 0] * (sx - lsx) + Jm[0:3, 1] * (sy - lsy) + Jm[0:3, 2] * (sw - lsw)
    Rcq = R(dq / np.linalg.norm(dq)) @ R(synth.so3_exp(Jm[3:6, 2] * (sw - lsw)))
    dtd = snap["td_wheel"] - ltd
    Ef, Eb = R(synth.so3_exp(sw * lin_gyr * dtd)), R(synth.so3_exp(-sw * gyr_1 * dtd))
    Rt = Ef @ Rcq @ Eb
    pt = Ef @ (sv @ lin_vel * dtd + cp - Rcq @ sv @ vel_1 * dtd)
    rp = (Ri @ Rio).T @ (Rj @ tio + Pj - Ri @ tio - Pi) - pt
    Rres = Rt.T @ (Ri @ Rio).T @ Rj @ Rio
    rq = synth.so3_log(synth.rot2q(Rres))
    return np.concatenate([rp, rq]), cov


@pytest.fixture(scope="module")
def snap():
    scn = synth.Scenario(seed=21, n_landmarks=60, use_wheel=True)
    s = scn.window(0)
    s["ix_wheel"] = np.array([1.01, 0.99, 1.015])
    s["td"], s["td_wheel"] = 0.002, 0.003
    return s


def test_numpy_visual(oracle, snap):
    ev = oracle.eval_factors(snap)
    mine = np.array([np_visual(snap, k) for k in range(len(snap["vis_imu_i"]))])
    np.testing.assert_allclose(ev["vis_r"], mine, rtol=0, atol=1e-10 * max(1, np.abs(mine).max()))


def test_numpy_imu(oracle, snap):
    ev = oracle.eval_factors(snap)
    for k in range(len(snap["imu_frame"])):
        raw, cov = np_imu_raw(snap, k)
        L = np.linalg.cholesky(np.linalg.inv(cov))
        want = L.T @ raw
        # sqrt_info comes from inverting a covariance with condition ~1e12: compare the
        # information-weighted norm tightly and the vector loosely
        assert abs(ev["imu_r"][k] @ ev["imu_r"][k] - raw @ np.linalg.solve(cov, raw)) < 1e-6 * (want @ want)
        np.testing.assert_allclose(ev["imu_r"][k], want, rtol=0, atol=1e-5 * np.abs(want).max())


def test_numpy_wheel(oracle, snap):
    ev = oracle.eval_factors(snap)
    for k in range(len(snap["wheel_frame"])):
        raw, cov = np_wheel_raw(snap, k)
        want = np.linalg.cholesky(np.linalg.inv(cov)).T @ raw
        np.testing.assert_allclose(ev["wheel_r"][k], want, rtol=0, atol=1e-8 * np.abs(want).max())


def test_sqrt_info_identity(oracle, snap):
    """sqrt_info^T sqrt_info == cov^-1 (imu_factor.h:73)."""
    cov = snap["imu"][0][17 + 225:].reshape(15, 15)
    S = oracle.sqrt_info(cov)
    assert np.allclose(np.triu(S), S)
    M = S.T @ S @ cov
    assert np.abs(M - np.eye(15)).max() < 1e-6


def test_preintegration_matches_numpy(oracle):
    scn = synth.Scenario(seed=3, n_landmarks=5, use_wheel=True)
    got = oracle.preintegrate_imu(scn.imu_raw[:4], scn.ba_est, scn.bg_est, [synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W])
    want = np.array(scn.imu_rec[:4])
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-16)
    gotw = oracle.preintegrate_wheel(scn.wheel_raw[:4], [1.0, 1.0, 1.0, 0.0], [synth.VEL_N_WHEEL, synth.GYR_N_WHEEL])
    wantw = np.array(scn.wheel_rec[:4])
    np.testing.assert_allclose(gotw, wantw, rtol=1e-9, atol=1e-16)


def test_preintegration_predicts_truth():
    """Noise-free mid-point pre-integration reproduces the true relative motion (sanity of the
    generator + of the IMU residual definition, integration_base.h:186-193)."""
    scn = synth.Scenario(seed=4, n_landmarks=5, use_wheel=True, noise=False)
    s = scn.window(0, state=scn.truth_state(0))
    s["speed_bias"][:, 3:6], s["speed_bias"][:, 6:9] = scn.ba_est, scn.bg_est
    for k in range(10):
        raw, _ = np_imu_raw(s, k)
        assert np.abs(raw[:9]).max() < 5e-6
        rw, _ = np_wheel_raw(s, k)
        assert np.abs(rw).max() < 5e-6


def test_sym_eig_against_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 40, 120):
        B = rng.normal(size=(n, n))
        A = B @ B.T * np.logspace(0, -6, n)[None, :]
        A = 0.5 * (A + A.T)
        w, V = oracle.sym_eig(A)
        np.testing.assert_allclose(w, np.linalg.eigvalsh(A), rtol=0, atol=1e-11 * np.abs(w).max())
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-11
        assert np.abs(V @ np.diag(w) @ V.T - A).max() < 1e-11 * max(1.0, np.abs(A).max())


# ---------------------------------------------------------------------------------------------
# a12: the trust-region loop itself against an independent implementation (tests/ceres_trust_region_np.py, written from
# Ceres 1.14's published algorithm on the stacked Jacobian: no Schur complement, no normal-equation partials).
# ---------------------------------------------------------------------------------------------
def _dogleg_fixture():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dogleg_np.npz"))


def _check_against_loop(summary, ref, name, final_rtol=1e-9):
    """accept / reject sequence, termination: identical. Costs: 1e-6 relative on the transient iterations (the first steps drop
    the cost by four orders of magnitude through an ill-conditioned system), 1e-9 on the final one of a run that settles
    (free_masks stops on rejected steps three accepted steps after a cost of 1e8: 1e-6 there too). Final radius: 1e-6."""
    n = len(ref[name + "_accepted"])
    assert summary["accepted"] == ref[name + "_accepted"].tolist()
    assert summary["iterations"] == n - 1 and summary["termination"] == int(ref[name + "_termination"])
    np.testing.assert_allclose(summary["cost_history"], ref[name + "_cost_history"], rtol=1e-6)
    assert abs(summary["final_cost"] - ref[name + "_cost_history"][-1]) < final_rtol * summary["final_cost"]
    assert abs(summary["final_radius"] - ref[name + "_radius_history"][-1]) < 1e-6 * summary["final_radius"]


@pytest.mark.parametrize("name", ["A", "B", "cfg1", "free_masks", "retry2"])
def test_dogleg_loop_oracle_reproduces_independent_numpy_loop(oracle, name):
    from dogleg_cases import cases
    ref = _dogleg_fixture()
    snap, kw = [(s, k) for n, s, k in cases(oracle) if n == name][0]
    orc = oracle.with_options(test_fail_chol_iter=kw["fail_chol_iter"]) if kw.get("fail_chol_iter") else oracle
    _check_against_loop(orc.solve(snap, abi.MARGIN_NONE)["summary"], ref, name, final_rtol=1e-6 if name == "free_masks" else 1e-9)


@pytest.mark.parametrize("name", ["B", "free_masks"])
def test_dogleg_loop_fixture_is_what_the_numpy_loop_produces(oracle, name):
    """The committed fixture is regenerated (tests/golden/make_golden_dogleg.py) and compared: radii and damping included."""
    import ceres_trust_region_np as ctr
    from dogleg_cases import cases
    ref = _dogleg_fixture()
    snap, kw = [(s, k) for n, s, k in cases(oracle) if n == name][0]
    r = ctr.solve(oracle, snap, **kw)
    assert r["accepted"] == ref[name + "_accepted"].tolist()
    np.testing.assert_allclose(r["cost_history"], ref[name + "_cost_history"], rtol=1e-9)
    np.testing.assert_allclose(r["radius_history"], ref[name + "_radius_history"], rtol=1e-9)
    np.testing.assert_allclose(r["mu_history"], ref[name + "_mu_history"], rtol=1e-12)
    if name == "free_masks":      # this case is in the set because it rejects steps: radius halvings with the linearisation kept
        assert 0 in r["accepted"][1:] and min(r["radius_history"]) < 1e4


def test_divide_and_conquer_tridiagonal():
    """tests/dc_eig_np.py — the numpy statement of the device's divide & conquer eigensolver (tridiag_dc, gfbe_marg.hip) — against
    numpy.linalg.eigh: residual |T V - V L|, orthogonality |V^T V - I| and the eigenvalues to a few eps |T| on random tridiagonal
    matrices of every size class, the tridiagonal form of 1e16-conditioned SPD matrices with a near-null space (what the marginalisation's
    A' looks like), matrices that split (zero off-diagonals), clustered and multiple eigenvalues, Wilkinson's W21+."""
    import dc_eig_np as dc
    import scipy.linalg as sl
    rng = np.random.default_rng(5)

    def check(d, e):
        n = len(d)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        lam, V = dc.dc_eig(np.asarray(d, float), np.asarray(e, float))
        nrm = max(np.abs(T).max(), 1e-300)
        assert np.abs(T @ V - V * lam[None, :]).max() <= 2e-14 * nrm
        assert np.abs(V.T @ V - np.eye(n)).max() <= 2e-14
        assert np.abs(np.sort(lam) - np.linalg.eigvalsh(T)).max() <= 2e-14 * nrm

    for n in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 43, 86, 87, 90):
        check(rng.normal(size=n), rng.normal(size=n - 1))
    for n in (61, 86):
        for _ in range(4):
            U = np.linalg.qr(rng.normal(size=(n, n)))[0]
            s = 10.0 ** rng.uniform(-9, 7, n)
            s[:4] = 1e-10 * rng.uniform(0.01, 1, 4)
            A = (U * s) @ U.T
            H = sl.hessenberg(0.5 * (A + A.T))
            check(np.diag(H).copy(), np.diag(H, -1).copy())
    n = 86
    check(np.ones(n), np.zeros(n - 1))
    check(2 * np.ones(n), -np.ones(n - 1))
    check(np.ones(n), 1e-9 * np.ones(n - 1))
    e = rng.normal(size=n - 1)
    e[::7] = 0.0
    check(rng.normal(size=n), e)
    check(np.abs(np.arange(-10, 11)).astype(float), np.ones(20))
    check(np.concatenate([np.linspace(1, 2, 40), np.linspace(1, 2, 40) + 1e-13]), 1e-7 * rng.normal(size=79))
