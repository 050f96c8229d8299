"""The optional in-window factors on the device (csrc/gfbe_optional.hip) vs the CPU oracle through the C ABI (SURVEY.md §8f
rank 2): same formulas, FP64, no fused multiply-add in either build -> 1e-13 relative; 11 factors (one window) and 20 000."""
import numpy as np
import pytest

from _gfbe_import import gf
from test_optional_oracle import NOISE_INV, plane_case, pose_plus, rq

abi = gf.abi
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


@pytest.mark.parametrize("n", [11, 20000])
def test_plane_factor_matches_oracle(be, oracle, n):
    pose, ex, q, z = plane_case(7, n)
    a = abi.plane_eval(oracle.lib, "gfo_", None, pose, ex, q, z, NOISE_INV)
    b = abi.plane_eval(be.lib, "gfbe_", be.ctx, pose, ex, q, z, NOISE_INV)
    scale = NOISE_INV.max() * (1 + np.abs(pose[:, :3]).max())
    assert np.abs(a["r"] - b["r"]).max() < 1e-13 * scale
    assert np.abs(a["J"] - b["J"]).max() < 1e-13 * scale
    assert abs(a["cost"] - b["cost"]) < 1e-12 * a["cost"]
    again = abi.plane_eval(be.lib, "gfbe_", be.ctx, pose, ex, q, z, NOISE_INV)
    assert again["cost"] == b["cost"] and np.array_equal(again["J"], b["J"])        # deterministic


@pytest.mark.parametrize("n", [1, 5000])
def test_pose_anchor_matches_oracle(be, oracle, n):
    rng = np.random.default_rng(9)
    anchor = np.array([np.concatenate([rng.normal(0, 2, 3), rq(rng)]) for _ in range(n)])
    pose = np.array([pose_plus(x, rng.normal(0, 0.05, 6)) for x in anchor])
    a = abi.anchor_eval(oracle.lib, "gfo_", None, pose, anchor, 120.0)
    b = abi.anchor_eval(be.lib, "gfbe_", be.ctx, pose, anchor, 120.0)
    assert np.abs(a["r"] - b["r"]).max() < 1e-12 and np.abs(a["J"] - b["J"]).max() < 1e-12
    assert abs(a["cost"] - b["cost"]) < 1e-12 * a["cost"]


def test_orientation_subset_plus_and_empty_inputs(be, oracle):
    rng = np.random.default_rng(2)
    for _ in range(20):
        q, d = rq(rng), rng.normal(0, 0.2, 3)
        np.testing.assert_allclose(abi.orientation_subset_plus(be.lib, "gfbe_", q, d), abi.orientation_subset_plus(oracle.lib, "gfo_", q, d), atol=1e-15)
    out = abi.plane_eval(be.lib, "gfbe_", be.ctx, np.zeros((0, 7)), [0, 0, 0, 0, 0, 0, 1.0], [0, 0, 0, 1.0], 0.0, NOISE_INV)
    assert out["cost"] == 0.0 and out["r"].shape == (0, 3)
