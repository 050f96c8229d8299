"""The GNSS factors on the device (csrc/gfbe_gnss.hip) vs the CPU oracle through the C ABI.
Pseudo-ranges are ~2.5e7 m in doubles (4e-9 m resolution) and weights reach 250, so residuals agree to ~1e-5 absolute;
Jacobians to 1e-9 relative; the clock factors are linear and agree to roundoff."""
import numpy as np
import pytest

from _gfbe_import import gf
import gnss_cases as gc

abi = gf.abi
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


@pytest.mark.parametrize("seed,lat,lon,h,n_per_frame", [(21, 22.3, 114.17, 30.0, 8), (22, -33.9, 151.2, 900.0, 3), (23, 68.0, -20.0, 5.0, 40)])
def test_gnss_factors_match_oracle(be, oracle, seed, lat, lon, h, n_per_frame):
    c = gc.gnss_case(seed, n_per_frame=n_per_frame, lat=lat, lon=lon, h=h)
    for iono in ("case", None):
        a = gc.eval_case(abi, oracle.lib, "gfo_", None, c, iono=iono)
        b = gc.eval_case(abi, be.lib, "gfbe_", be.ctx, c, iono=iono)
        assert np.abs(a["r"] - b["r"]).max() < 1e-5
        assert np.abs(a["J"] - b["J"]).max() < 1e-9 * np.abs(a["J"]).max()
        np.testing.assert_allclose(b["r_dt_ddt"], a["r_dt_ddt"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(b["r_smooth"], a["r_smooth"], rtol=0, atol=1e-12)
        assert abs(a["cost"] - b["cost"]) < 1e-9 * a["cost"]
    again = gc.eval_case(abi, be.lib, "gfbe_", be.ctx, c)
    first = gc.eval_case(abi, be.lib, "gfbe_", be.ctx, c)
    assert again["cost"] == first["cost"] and np.array_equal(again["J"], first["J"])        # deterministic


def test_gnss_edge_cases(be, oracle):
    c = gc.gnss_case(24, n_per_frame=1)
    # no observations: the clock chain alone
    empty = dict(c, obs=[])
    a, b = gc.eval_case(abi, oracle.lib, "gfo_", None, empty), gc.eval_case(abi, be.lib, "gfbe_", be.ctx, empty)
    assert b["r"].shape == (0, 2) and abs(a["cost"] - b["cost"]) < 1e-12 * a["cost"] and a["cost"] > 0
    # a satellite below the horizon: no atmosphere terms, finite residuals, same as the oracle
    e, n, u = gc.enu_axes(22.3, 114.17)
    c["obs"][0]["sv_pos"] = c["anc"] + 2.0e7 * (np.cos(-0.2) * n + np.sin(-0.2) * u)
    a, b = gc.eval_case(abi, oracle.lib, "gfo_", None, c), gc.eval_case(abi, be.lib, "gfbe_", be.ctx, c)
    assert np.isfinite(b["r"]).all() and np.abs(a["r"] - b["r"]).max() < 1e-5
    # indices out of range are refused before anything is launched
    bad = dict(c, obs=[dict(c["obs"][0], lower_idx=10)])
    with pytest.raises(RuntimeError):
        gc.eval_case(abi, be.lib, "gfbe_", be.ctx, bad)
    bad = dict(c, obs=[dict(c["obs"][0], pr_uura=0.0)])
    with pytest.raises(RuntimeError):
        gc.eval_case(abi, be.lib, "gfbe_", be.ctx, bad)
