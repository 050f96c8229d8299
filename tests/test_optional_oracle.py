"""CPU pins of the optional in-window factors (oracle/gfo_optional.cpp): an independent numpy
restatement of the residual equations (plane_factor.h:45-47, pose_anchor_factor.cpp:12-16), central differences on the
manifold (the protocol of projectionTwoFrameOneCamFactor.cpp:214-274: right perturbation q * deltaQ(d)), and a hand-computed
case. The reference holds no known-answer test for these factors (parity unpinned)."""
import numpy as np

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
NOISE_INV = np.array([1 / 0.01, 1 / 0.02, 1 / 0.05])


def rq(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def pose_plus(x, d6):
    dq = np.concatenate([0.5 * d6[3:], [1.0]])
    q = synth.qmul(x[3:], dq / np.linalg.norm(dq))
    return np.concatenate([x[:3] + d6[:3], q / np.linalg.norm(q)])


def plane_residual(pose, ex, q, z):
    Ri, Rio, Rpw = synth.qrot(pose[3:]), synth.qrot(ex[3:]), synth.qrot(q)
    up = Rio.T @ Ri.T @ Rpw.T @ np.array([0, 0, 1.0])
    return NOISE_INV * np.array([up[0], up[1], z + (Rpw @ (pose[:3] + Ri @ ex[:3]))[2]])


def plane_case(seed, n=5):
    rng = np.random.default_rng(seed)
    pose = np.array([np.concatenate([rng.normal(0, 2, 3), rq(rng)]) for _ in range(n)])
    ex = np.concatenate([rng.normal(0, 0.3, 3), rq(rng)])
    return pose, ex, rq(rng), rng.normal()


def test_plane_factor_residual_and_jacobians(oracle):
    pose, ex, q, z = plane_case(1)
    out = abi.plane_eval(oracle.lib, "gfo_", None, pose, ex, q, z, NOISE_INV)
    ref = np.array([plane_residual(p, ex, q, z) for p in pose])
    np.testing.assert_allclose(out["r"], ref, rtol=0, atol=1e-12 * NOISE_INV.max())
    assert abs(out["cost"] - 0.5 * (ref ** 2).sum()) < 1e-10 * out["cost"]
    h = 1e-6
    for k in range(len(pose)):
        num = np.zeros((3, 16))
        for c in range(16):
            d = np.zeros(16)
            d[c] = h
            def at(sgn):
                dd = sgn * d
                qq = abi.orientation_subset_plus(oracle.lib, "gfo_", q, dd[12:15], constant=(0, 0, 0))
                return plane_residual(pose_plus(pose[k], dd[:6]), pose_plus(ex, dd[6:12]), qq, z + dd[15])
            num[:, c] = (at(1) - at(-1)) / (2 * h)
        assert np.abs(out["J"][k] - num).max() < 1e-6 * NOISE_INV.max()
    # structure: roll / pitch rows do not see positions or the height; the height row has d/dz = ZPW_N_INV
    assert not out["J"][:, :2, [0, 1, 2, 6, 7, 8, 15]].any()
    np.testing.assert_array_equal(out["J"][:, 2, 15], NOISE_INV[2])


def test_plane_factor_level_ground_by_hand(oracle):
    """Robot standing on the plane z = -0.3 with the odometer 0.3 below the IMU: zero residual; a 0.01 rad roll of the body gives
    (up_o)_y = sin(0.01) in the second component, a lift of 5 cm gives 0.05 * ZPW_N_INV in the third."""
    ex = np.array([0, 0, -0.3, 0, 0, 0, 1.0])
    ident = np.array([0, 0, 0, 1.0])
    out = abi.plane_eval(oracle.lib, "gfo_", None, [[1.0, 2.0, 0.0, 0, 0, 0, 1]], ex, ident, 0.3, NOISE_INV)
    np.testing.assert_allclose(out["r"], 0, atol=1e-13)
    a = 0.01
    rolled = np.array([1.0, 2.0, 0.05, np.sin(a / 2), 0, 0, np.cos(a / 2)])
    out = abi.plane_eval(oracle.lib, "gfo_", None, [rolled], np.array([0, 0, 0, 0, 0, 0, 1.0]), ident, 0.0, NOISE_INV)
    np.testing.assert_allclose(out["r"][0], [0.0, NOISE_INV[1] * np.sin(a), NOISE_INV[2] * 0.05], atol=1e-12)


def anchor_residual(x, a, s):
    qa = a[3:]
    qa_inv = np.concatenate([-qa[:3], qa[3:]]) / (qa @ qa)
    return s * np.concatenate([x[:3] - a[:3], 2.0 * synth.qmul(x[3:], qa_inv)[:3]])


def test_pose_anchor_factor(oracle):
    rng = np.random.default_rng(3)
    anchor = np.array([np.concatenate([rng.normal(0, 2, 3), rq(rng)]) for _ in range(4)])
    pose = np.array([pose_plus(a, rng.normal(0, 0.05, 6)) for a in anchor])
    out = abi.anchor_eval(oracle.lib, "gfo_", None, pose, anchor, 120.0)
    ref = np.array([anchor_residual(x, a, 120.0) for x, a in zip(pose, anchor)])
    np.testing.assert_allclose(out["r"], ref, atol=1e-11)
    assert abs(out["cost"] - 0.5 * (ref ** 2).sum()) < 1e-10 * out["cost"]
    # At the anchor itself the rotation block equals the true derivative of 2 vec(q * q_a^-1) wrt the right perturbation only up
    # to the rotation R(q_a): d/dtheta = Qleft(q)[br] Qright(q_a^-1)[br] ~ R-dependent, while the reference keeps Qright(q_a^-1) alone
    # and doubles everything: its position block is 2 * sqrt_info * I for a residual sqrt_info * (p - p_a). Pin both facts.
    for k in range(4):
        np.testing.assert_allclose(out["J"][k][:3, :3], 240.0 * np.eye(3), atol=1e-12)
        assert not out["J"][k][:3, 3:].any() and not out["J"][k][3:, :3].any()
        qa = anchor[k, 3:]
        w, x, y, z = qa[3], -qa[0], -qa[1], -qa[2]
        np.testing.assert_allclose(out["J"][k][3:, 3:], 240.0 * np.array([[w, z, -y], [-z, w, x], [y, -x, w]]), atol=1e-12)
    h, x0 = 1e-6, anchor[0]
    num = np.zeros((6, 6))
    for c in range(6):
        d = np.zeros(6)
        d[c] = h
        num[:, c] = (anchor_residual(pose_plus(x0, d), x0, 120.0) - anchor_residual(pose_plus(x0, -d), x0, 120.0)) / (2 * h)
    at_anchor = abi.anchor_eval(oracle.lib, "gfo_", None, [x0], [x0], 120.0)
    np.testing.assert_allclose(at_anchor["r"], 0, atol=1e-12)
    np.testing.assert_allclose(num[:3, :3], 120.0 * np.eye(3), atol=1e-6)             # the true position derivative: half the reference's
    np.testing.assert_allclose(at_anchor["J"][0][:3, :3], 2 * num[:3, :3], atol=1e-6)


def test_orientation_subset_plus(oracle):
    rng = np.random.default_rng(5)
    q, d = rq(rng), rng.normal(0, 0.1, 3)
    full = abi.orientation_subset_plus(oracle.lib, "gfo_", q, d, constant=(0, 0, 0))
    dq = np.concatenate([0.5 * d, [1.0]])
    ref = synth.qmul(q, dq / np.linalg.norm(dq))
    np.testing.assert_allclose(full, ref / np.linalg.norm(ref), atol=1e-15)
    masked = abi.orientation_subset_plus(oracle.lib, "gfo_", q, d, constant=(0, 0, 1))       # the plane's yaw is not observable
    np.testing.assert_allclose(masked, abi.orientation_subset_plus(oracle.lib, "gfo_", q, [d[0], d[1], 0.0], constant=(0, 0, 0)), atol=0)
    assert abs(np.linalg.norm(masked) - 1) < 1e-15
