"""CPU pins of the joint LIO + VIO window of the oracle (BASELINE configs[4]; SURVEY.md §8f rank 4: LiDAR point-to-plane
factors on the newest pose, a capability the reference does not have — there is nothing in it to compare with, so the
oracle is pinned by numpy: the extra cost / gradient / Gauss-Newton block equal a direct restatement of
LidarPlaneNormFactor (lidarFactor.cpp:18-51) under HuberLoss (lidarodom.cpp:539), and the solve moves the newest pose towards
the truth when the scan is exact."""
import numpy as np

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth


def numpy_lidar_terms(snap, lio):
    x = snap["pose"][lio["frame"]]
    R, t = synth.qrot(x[3:]), x[:3]
    sw = lio["sqrt_info"] * lio["weights"]
    r = sw * ((lio["normals"] * (lio["pts"] @ R.T + t)).sum(axis=1) + lio["offsets"])
    nR = lio["normals"] @ R
    J = np.concatenate([sw[:, None] * lio["normals"], -sw[:, None] * np.cross(nR, lio["pts"])], axis=1)
    d = lio["huber_delta"]
    s = r * r
    rho = np.where(s <= d * d, s, 2 * d * np.sqrt(s) - d * d)
    scale = np.where(s <= d * d, 1.0, np.sqrt(d / np.sqrt(np.maximum(s, 1e-300))))
    Jc, rc = J * scale[:, None], r * scale
    return 0.5 * rho.sum(), Jc.T @ rc, Jc.T @ Jc


def test_linearisation_adds_exactly_the_lidar_terms(oracle):
    scn = synth.Scenario(seed=4, n_landmarks=150, use_wheel=True)
    snap = scn.window(0)
    lio = synth.lidar_block(scn, 0, n=400, seed=1, outliers=0.1)
    base = oracle.linearize(snap)
    with_lio = oracle.linearize(dict(snap, lio=lio))
    cost, g, H = numpy_lidar_terms(snap, lio)
    assert (np.abs(lio["sqrt_info"] * lio["weights"] * 1.0) > 0).all() and cost > 0
    o = 6 * lio["frame"]
    assert abs((with_lio["cost"] - base["cost"]) - cost) < 1e-9 * cost
    dH, dg = with_lio["H"] - base["H"], with_lio["g"] - base["g"]
    np.testing.assert_allclose(dH[o:o + 6, o:o + 6], H, rtol=1e-9, atol=1e-9 * np.abs(H).max())
    np.testing.assert_allclose(dg[o:o + 6], g, rtol=1e-9, atol=1e-9 * np.abs(g).max())
    dH[o:o + 6, o:o + 6] = 0
    dg[o:o + 6] = 0
    assert np.abs(dH).max() < 1e-9 * np.abs(base["H"]).max() and np.abs(dg).max() < 1e-9 * np.abs(base["g"]).max()
    # some residuals sit on the linear branch of the Huber loss
    x = snap["pose"][lio["frame"]]
    r = lio["sqrt_info"] * lio["weights"] * ((lio["normals"] * (lio["pts"] @ synth.qrot(x[3:]).T + x[:3])).sum(axis=1) + lio["offsets"])
    assert (np.abs(r) > lio["huber_delta"]).sum() > 10 and (np.abs(r) < lio["huber_delta"]).sum() > 10


def test_exact_scan_pulls_the_newest_pose_to_the_truth(oracle):
    scn = synth.Scenario(seed=6, n_landmarks=200, use_wheel=True)
    snap = scn.window(0)
    truth = scn.truth_state(0)["pose"][abi.WINDOW_SIZE]
    lio = synth.lidar_block(scn, 0, n=2000, seed=2, noise=0.0, sqrt_info=200.0)
    plain = oracle.solve(snap, abi.MARGIN_NONE)
    joint = oracle.solve(dict(snap, lio=lio), abi.MARGIN_NONE)
    # the re-anchoring of optimization() pins position and yaw of pose 0, so compare the newest pose RELATIVE to pose 0
    def rel(st):
        R0 = synth.qrot(st["pose"][0, 3:])
        return R0.T @ (st["pose"][abi.WINDOW_SIZE, :3] - st["pose"][0, :3])
    t_state = scn.truth_state(0)
    e_plain = np.linalg.norm(rel(plain["state"]) - rel(t_state))
    e_joint = np.linalg.norm(rel(joint["state"]) - rel(t_state))
    assert e_joint < e_plain
    assert joint["summary"]["final_cost"] < joint["summary"]["initial_cost"]
    assert truth.shape == (7,)


def test_bad_lidar_frame_is_rejected(oracle):
    import pytest
    scn = synth.Scenario(seed=4, n_landmarks=50, use_wheel=False)
    lio = synth.lidar_block(scn, 0, n=10, frame=3)
    lio["frame"] = 12
    with pytest.raises(Exception):
        oracle.solve(dict(scn.window(0), lio=lio), abi.MARGIN_NONE)
