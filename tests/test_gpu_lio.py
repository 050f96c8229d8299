"""LiDAR point-to-plane factors on the device (csrc/gfbe_lio.hip) vs the CPU oracle through the C ABI:
residuals / Jacobians 1e-12 relative, the reduced normal equations 1e-11 (summation order), on scans of 2 000
(max_num_residuals, lio/config/m3dgr.yaml:45) and 100 000 residuals, plus the reference's known input."""
import numpy as np
import pytest

from _gfbe_import import gf
from test_lio_oracle import scan

abi = gf.abi
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


@pytest.mark.parametrize("ct", [0, 1])
@pytest.mark.parametrize("n", [1, 2000, 100000])
def test_linearize_matches_oracle(be, oracle, ct, n):
    pts, normals, offs, alpha, w, pb, pe = scan(np.random.default_rng(7 + n), n, bool(ct))
    a = abi.lio_linearize(oracle.lib, "gfo_", None, ct, pts, normals, offs, alpha, w, 0.8, pb, pe)
    b = abi.lio_linearize(be.lib, "gfbe_", be.ctx, ct, pts, normals, offs, alpha, w, 0.8, pb, pe)
    assert np.abs(a["r"] - b["r"]).max() <= 1e-12 * max(1.0, np.abs(a["r"]).max())
    assert np.abs(a["J"] - b["J"]).max() <= 1e-12 * max(1.0, np.abs(a["J"]).max())
    assert np.abs(a["H"] - b["H"]).max() <= 1e-11 * np.abs(a["H"]).max()
    assert np.abs(a["g"] - b["g"]).max() <= 1e-11 * np.abs(a["g"]).max()
    assert abs(a["cost"] - b["cost"]) <= 1e-12 * a["cost"]
    again = abi.lio_linearize(be.lib, "gfbe_", be.ctx, ct, pts, normals, offs, alpha, w, 0.8, pb, pe)
    assert np.array_equal(again["H"], b["H"]) and again["cost"] == b["cost"]       # fixed-order reduction


def test_reference_known_input_on_device(be):
    normal = np.array([0.3, 1.5, -2.0])
    normal /= np.linalg.norm(normal)
    q = np.array([0.6, 1.3, -0.9, 0.2])
    pose = np.concatenate([[11.0, 13, 15], q / np.linalg.norm(q)])
    out = abi.lio_linearize(be.lib, "gfbe_", be.ctx, 0, [[10.0, 12, 14]], [normal], [-normal @ np.array([1.0, 3, 5])], None, [1.0], 1.0, pose)
    R = gf.synth.qrot(pose[3:])
    want_r = (R @ np.array([10.0, 12, 14]) + pose[:3] - np.array([1.0, 3, 5])) @ normal
    assert abs(out["r"][0] - want_r) < 1e-12
    np.testing.assert_allclose(out["J"][0, :3], normal, atol=1e-15)
