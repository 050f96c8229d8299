"""CPU pins of the pose-graph oracle (oracle/gfo_posegraph.cpp): analytic tangent Jacobians of
RelativeRTError vs central differences through ceres::QuaternionParameterization::Plus (the reference differentiates
automatically: Factors.h:102-109), an independent dense numpy Levenberg-Marquardt step, and convergence on the
BASELINE configs[3] graph."""
import numpy as np

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth


def small_graph(seed=0, n=6):
    rng = np.random.default_rng(seed)
    pose, meas = np.zeros((n, 7)), np.zeros((n - 1, 7))
    for i in range(n):
        q = rng.normal(size=4)
        pose[i] = np.concatenate([rng.normal(size=3) * 2, q / np.linalg.norm(q)])
    for k in range(n - 1):
        q = rng.normal(size=4)
        meas[k] = np.concatenate([rng.normal(size=3), q / np.linalg.norm(q)])
    return dict(pose=pose, rel_i=np.arange(n - 1), rel_meas=meas, fix_i=[0, 3], fix_meas=[[0.1, 0.2, 0.3, 0.5], [5, 5, 5, 0.5]])


def test_jacobians_match_central_differences(oracle):
    pg = abi.PoseGraph(oracle.lib, "gfo_", None)
    g = small_graph()
    e = pg.eval(g)
    h, worst = 1e-6, 0.0
    for k in range(len(g["rel_i"])):
        for idx, col0 in ((k, 0), (k + 1, 6)):
            for c in range(6):
                d = np.zeros(6)
                d[c] = h
                plus, minus = g["pose"].copy(), g["pose"].copy()
                plus[idx], minus[idx] = abi.pg_plus(g["pose"][idx], d), abi.pg_plus(g["pose"][idx], -d)
                num = (pg.eval(dict(g, pose=plus))["rel_r"][k] - pg.eval(dict(g, pose=minus))["rel_r"][k]) / (2 * h)
                worst = max(worst, np.abs(num - e["rel_J"][k][:, col0 + c]).max() / max(1.0, np.abs(num).max()))
    assert worst < 1e-7, worst      # the reference's own Jacobian-check threshold is 1e-6 (LIO/apps/test_analytic_factor.cpp:134)


def test_first_lm_step_matches_dense_numpy(oracle):
    """One Levenberg-Marquardt iteration re-derived densely with numpy from the oracle's own residuals / Jacobians:
    Jacobi scaling, diagonal clamp(diag) / radius with radius 1e4, step, Plus — equals the oracle's first accepted iterate."""
    pg = abi.PoseGraph(oracle.lib, "gfo_", None)
    g = synth.pose_graph(n=40, seed=5, fix_every=5)
    e = pg.eval(g)
    n = len(g["pose"])
    rows = []
    J = np.zeros((6 * (n - 1) + 3 * len(g["fix_i"]), 6 * n))
    r = np.zeros(J.shape[0])
    for k, i in enumerate(g["rel_i"]):
        J[6 * k:6 * k + 6, 6 * i:6 * i + 12] = e["rel_J"][k]
        r[6 * k:6 * k + 6] = e["rel_r"][k]
    off = 6 * (n - 1)
    for k, i in enumerate(g["fix_i"]):
        rr = (g["pose"][i, :3] - g["fix_meas"][k, :3]) / g["fix_meas"][k, 3]
        sq = rr @ rr
        if sq > 1.0:      # Huber (delta = 1): rho'' <= 0, so the Corrector only rescales by sqrt(rho') (corrector.cc)
            rho1 = 1.0 / np.sqrt(sq)
            Jc = np.sqrt(rho1) * np.eye(3) / g["fix_meas"][k, 3]
            rr = np.sqrt(rho1) * rr
        else:
            Jc = np.eye(3) / g["fix_meas"][k, 3]
        J[off + 3 * k:off + 3 * k + 3, 6 * i + 3:6 * i + 6] = Jc
        r[off + 3 * k:off + 3 * k + 3] = rr
    np.testing.assert_allclose(r[off:], e["fix_r"].ravel(), rtol=1e-13, atol=1e-15)
    H, gr = J.T @ J, J.T @ r
    s = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    Hs, gs = H * np.outer(s, s), s * gr
    D2 = np.clip(np.diag(Hs), 1e-6, 1e32)
    y = np.linalg.solve(Hs + np.diag(D2 / 1e4), -gs)
    step = s * y
    want = np.array([abi.pg_plus(g["pose"][i], step[6 * i:6 * i + 6]) for i in range(n)])
    got = pg.solve(g, max_iterations=1)
    assert got["summary"]["accepted"] == [0, 1]
    np.testing.assert_allclose(got["pose"], want, rtol=0, atol=1e-10)


def test_config4_graph_converges_and_removes_drift(oracle):
    pg = abi.PoseGraph(oracle.lib, "gfo_", None)
    g = synth.pose_graph(n=5000)
    res = pg.solve(g, max_iterations=5)      # max_num_iterations = 5 (globalOpt.cpp:121)
    sm = res["summary"]
    assert sm["iterations"] <= 5 and sm["num_successful"] >= 3
    hist = np.array(sm["cost_history"])
    assert np.all(np.diff(hist[np.array(sm["accepted"], bool) | (np.arange(len(hist)) == 0)]) <= 0)     # accepted steps never increase the cost
    err0 = np.linalg.norm(g["pose"][:, :3] - g["truth"][:, :3], axis=1)
    err1 = np.linalg.norm(res["pose"][:, :3] - g["truth"][:, :3], axis=1)
    assert err1.mean() < 0.25 * err0.mean() and err1.mean() < 0.5          # drift of the dead-reckoned chain removed
    assert np.abs(np.linalg.norm(res["pose"][:, 3:], axis=1) - 1).max() < 1e-12   # Plus keeps the quaternions unit
