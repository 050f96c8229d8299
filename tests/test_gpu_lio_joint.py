"""Joint LIO + VIO window on the device vs the CPU oracle (BASELINE configs[4]: the cfg-2 window + 2000 point-to-plane
residuals on the newest pose). Same stated tolerances as the plain window solve (tests/test_gpu_parity.py::check_solve)."""
import numpy as np
import pytest

from _gfbe_import import gf
from test_gpu_parity import check_solve, window_with_prior

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


@pytest.mark.parametrize("flag", [abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW])
def test_joint_window_cfg5(be, oracle, flag):
    scn, snap = window_with_prior(oracle, 20250712, 2000)
    snap["lio"] = synth.lidar_block(scn, 1, n=2000, seed=3, outliers=0.05)
    want, got = check_solve(be, oracle, snap, flag)
    plain = be.solve({k: v for k, v in snap.items() if k != "lio"}, flag)
    assert np.abs(plain["state"]["pose"][abi.WINDOW_SIZE] - got["state"]["pose"][abi.WINDOW_SIZE]).max() > 1e-6     # the scan matters
    assert got["summary"]["initial_cost"] > plain["summary"]["initial_cost"]


def test_batch_with_and_without_scans_is_bit_identical_to_single_solves(be, oracle):
    """Three windows in one batch: no scan, 300 factors (fewer than one workgroup stride), 5000 factors on an inner pose."""
    scn = synth.Scenario(seed=77, n_landmarks=300, use_wheel=True)
    snaps = [scn.window(0), dict(scn.window(0), lio=synth.lidar_block(scn, 0, n=300, seed=5)),
             dict(scn.window(0), lio=synth.lidar_block(scn, 0, n=5000, seed=6, frame=7, huber_delta=0.0))]
    batch = be.solve_batch(snaps, abi.MARGIN_OLD)
    for snap, res in zip(snaps, batch):
        one = be.solve(snap, abi.MARGIN_OLD)
        assert one["summary"] == res["summary"]
        np.testing.assert_array_equal(one["state"]["pose"], res["state"]["pose"])
        np.testing.assert_array_equal(one["feature"], res["feature"])
    check_solve(be, oracle, snaps[2], abi.MARGIN_OLD)       # no loss (huber_delta = 0), inner pose
    assert batch[0]["summary"]["final_cost"] != batch[1]["summary"]["final_cost"]


def test_bad_lidar_block_fails_loudly(be):
    scn = synth.Scenario(seed=4, n_landmarks=50, use_wheel=False)
    lio = synth.lidar_block(scn, 0, n=10)
    lio["frame"] = 12
    with pytest.raises(gf.BackendError, match="lio"):
        be.solve(dict(scn.window(0), lio=lio), abi.MARGIN_NONE)
