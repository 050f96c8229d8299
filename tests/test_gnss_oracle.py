"""CPU pins of the GNSS factors (oracle/gfo_gnss.cpp): the independent numpy statement of
tests/gnss_cases.py, central differences against the analytic Jacobian (with the terms the reference's Jacobian leaves out held
fixed, and loosely with everything free), closed-form geodesy / zenith / night-time cases. The reference holds no known-answer
test for these factors and gnss_comm does not build here (ROS, glog): parity unpinned against the reference binary."""
import numpy as np

from _gfbe_import import gf
import gnss_cases as gc

abi = gf.abi


def test_residuals_match_the_numpy_statement(oracle):
    for seed, lat, lon, h in [(1, 22.3, 114.17, 30.0), (2, -33.9, 151.2, 900.0), (3, 68.0, -20.0, 5.0), (4, 4.0, 100.0, -20.0)]:
        c = gc.gnss_case(seed, lat=lat, lon=lon, h=h)
        out = gc.eval_case(abi, oracle.lib, "gfo_", None, c)
        ref = np.array([gc.psr_dopp_residual(o, c["iono"], c["pose"][o["lower_idx"], :3], c["speed_bias"][o["lower_idx"], :3],
                                             c["pose"][o["lower_idx"] + 1, :3], c["speed_bias"][o["lower_idx"] + 1, :3],
                                             c["rcv_dt"][o["frame"], o["sys_idx"]], c["rcv_ddt"][o["frame"]], c["yaw"], c["anc"])[0] for o in c["obs"]])
        # pseudo-ranges are ~2.5e7 m in doubles (4e-9 m resolution), weights up to 50
        assert np.abs(out["r"] - ref).max() < 2e-6, np.abs(out["r"] - ref).max()
        assert np.abs(out["r"]).max() < 300.0                      # the synthetic measurements carry metre-level noise only
        dd = np.array([[(c["rcv_dt"][i + 1, k] - c["rcv_dt"][i, k] - 0.5 * (c["rcv_ddt"][i] + c["rcv_ddt"][i + 1]) * c["frame_dt"][i]) * 50.0
                        for i in range(10)] for k in range(4)])
        np.testing.assert_allclose(out["r_dt_ddt"], dd, rtol=0, atol=1e-9)
        np.testing.assert_allclose(out["r_smooth"], (c["rcv_ddt"][:-1] - c["rcv_ddt"][1:]) * c["ddt_weight"], rtol=0, atol=1e-12)
        total = 0.5 * (out["r"] ** 2).sum() + 0.5 * (out["r_dt_ddt"] ** 2).sum() + 0.5 * (out["r_smooth"] ** 2).sum()
        assert abs(out["cost"] - total) < 1e-12 * total
        # no ionosphere parameters -> the Klobuchar term drops out (ion_delay stays 0)
        noi = gc.eval_case(abi, oracle.lib, "gfo_", None, c, iono=None)
        ref0 = np.array([gc.psr_dopp_residual(o, None, c["pose"][o["lower_idx"], :3], c["speed_bias"][o["lower_idx"], :3],
                                              c["pose"][o["lower_idx"] + 1, :3], c["speed_bias"][o["lower_idx"] + 1, :3],
                                              c["rcv_dt"][o["frame"], o["sys_idx"]], c["rcv_ddt"][o["frame"]], c["yaw"], c["anc"])[0] for o in c["obs"]])
        assert np.abs(noi["r"] - ref0).max() < 2e-6 and np.abs(noi["r"][:, 0] - out["r"][:, 0]).max() > 1e-3


def _numeric_jacobian(c, o, freeze):
    lo = o["lower_idx"]
    x0 = np.concatenate([c["pose"][lo, :3], c["speed_bias"][lo, :3], c["pose"][lo + 1, :3], c["speed_bias"][lo + 1, :3],
                         [c["rcv_dt"][o["frame"], o["sys_idx"]], c["rcv_ddt"][o["frame"]], c["yaw"]], c["anc"]])
    f = lambda x: gc.psr_dopp_residual(o, c["iono"], x[0:3], x[3:6], x[6:9], x[9:12], x[12], x[13], x[14], x[15:18], freeze=freeze)[0]
    steps = np.array([1e-1] * 3 + [1e-2] * 3 + [1e-1] * 3 + [1e-2] * 3 + [1e-1, 1e-3, 1e-4] + [1e-1] * 3)
    J = np.zeros((2, 18))
    for k in range(18):
        d = np.zeros(18)
        d[k] = steps[k]
        J[:, k] = (f(x0 + d) - f(x0 - d)) / (2 * steps[k])
    return J


def test_jacobian_against_central_differences(oracle):
    c = gc.gnss_case(5, n_per_frame=2)
    out = gc.eval_case(abi, oracle.lib, "gfo_", None, c)
    for k, o in enumerate(c["obs"]):
        lo = o["lower_idx"]
        _, nom = gc.psr_dopp_residual(o, c["iono"], c["pose"][lo, :3], c["speed_bias"][lo, :3], c["pose"][lo + 1, :3], c["speed_bias"][lo + 1, :3],
                                      c["rcv_dt"][o["frame"], o["sys_idx"]], c["rcv_ddt"][o["frame"]], c["yaw"], c["anc"])
        wp, wd = nom["sin2"] / o["pr_uura"] * 10.0, nom["sin2"] / o["dp_uura"] * 50.0
        J = out["J"][k]
        # (1) with the weights / atmosphere / Sagnac terms held at their nominal values the analytic Jacobian is exact in every
        #     column but anc_ecef, which the reference fills "for simplicity" (pseudo-range row only, frame rotation held)
        num = _numeric_jacobian(c, o, nom)
        assert np.abs(J[0, :14] - num[0, :14]).max() < 2e-5 * wp, (k, np.abs(J[0, :14] - num[0, :14]).max() / wp)
        assert abs(J[0, 14] - num[0, 14]) < 1e-5 * wp * (1 + np.abs(c["pose"][:, :3]).max())                 # yaw x local position
        assert np.abs(J[1, :14] - num[1, :14]).max() < 2e-5 * wd, (k, np.abs(J[1, :14] - num[1, :14]).max() / wd)
        # Doppler row, yaw column: the reference differentiates the velocity's rotation only; the line of sight also turns with
        # the receiver position (|sv_vel - V| / range x |local position| ~ 5e-3), which its Jacobian leaves out
        los_term = 4000.0 / 2.0e7 * (1 + np.abs(c["pose"][:, :3]).max())
        assert abs(J[1, 14] - num[1, 14]) < wd * (1e-4 * (1 + np.abs(c["speed_bias"][:, :3]).max()) + los_term)
        assert np.abs(J[0, 15:] - num[0, 15:]).max() < 1e-3 * wp and not J[1, 15:].any()
        # (2) with everything free the neglected terms show up at the 1e-2 level at most (the troposphere's height gradient
        #     through a low-elevation mapping function is the largest)
        full = _numeric_jacobian(c, o, None)
        assert np.abs(J[0, :14] - full[0, :14]).max() < 2e-2 * wp
        assert np.abs(J[1, :14] - full[1, :14]).max() < 2e-2 * wd
        # structure: the pseudo-range row does not see velocities or the clock drift, the Doppler row does not see the clock bias
        assert not J[0, [3, 4, 5, 9, 10, 11, 13]].any() and J[1, 12] == 0.0
        assert J[0, 12] == wp or abs(J[0, 12] - wp) < 1e-9 * wp
        assert abs(J[1, 13] - wd) < 1e-9 * wd


def test_closed_form_cases(oracle):
    c = gc.gnss_case(6, n_per_frame=1)
    # a satellite straight above the receiver: elevation 90 degrees -> weights 10 / uura and 50 / uura, mapping functions 1
    o = c["obs"][0]
    lo = o["lower_idx"]
    e, n, u = gc.enu_axes(22.3, 114.17)
    c["pose"][:, :3] = 0.0
    c["speed_bias"][:, :3] = 0.0
    o.update(sv_pos=c["anc"] + 2.0e7 * u, sv_vel=np.zeros(3), svdt=0.0, svddt=0.0, tgd=0.0, psr=0.0, dopp=0.0, pr_uura=2.0, dp_uura=0.5, doy=28.0)
    c["rcv_dt"][:] = 0.0
    c["rcv_ddt"][:] = 0.0
    c["obs"] = [o]
    out = gc.eval_case(abi, oracle.lib, "gfo_", None, c, iono=None)
    lla = gc.ecef2geo_iter(c["anc"])
    zenith = gc.trop(28.0, lla, np.pi / 2)
    assert 2.2 < zenith < 2.6                                    # Saastamoinen at sea level: ~2.3 m dry + ~0.1-0.3 m wet
    sag = gc.OMEGA_E * (o["sv_pos"][0] * c["anc"][1] - o["sv_pos"][1] * c["anc"][0]) / gc.C_LIGHT
    # geodetic 'up' through a point at height 30 m: the range is 2e7 to sub-millimetre
    assert abs(out["r"][0, 0] - (2.0e7 + sag + zenith) * 10.0 / 2.0) < 1e-4
    assert abs(out["r"][0, 1]) < 1e-9
    assert abs(out["J"][0, 0, 12] - 5.0) < 1e-9 and abs(out["J"][0, 1, 13] - 100.0) < 1e-7
    # night-time Klobuchar: |x| >= 1.57 -> the constant 5 ns floor times the obliquity factor
    az, el, lla = gc.azel(c["anc"], o["sv_pos"])
    o["tow"] = ((0.0 - 43200.0 * (lla[1] / 180.0)) % 86400.0) + 3600.0 * 2          # 02:00 local time
    with_ion = gc.eval_case(abi, oracle.lib, "gfo_", None, c)
    f = 1 + 16 * (0.53 - 0.5) ** 3
    assert abs((with_ion["r"][0, 0] - out["r"][0, 0]) / 5.0 - gc.C_LIGHT * f * 5e-9) < 1e-6
    # geodesy: the closed form against the fixed-point iteration through the factor's frame (anchor on the equator / near a pole)
    for lat, lon, h in [(0.0, 0.0, 0.0), (89.9, 40.0, 100.0), (-45.0, -170.0, 3000.0)]:
        anc = gc.geo2ecef(lat, lon, h)
        np.testing.assert_allclose(gc.ecef2geo_iter(anc), [lat, lon, h], atol=1e-8)
        c2 = gc.gnss_case(7, n_per_frame=1, lat=lat, lon=lon, h=h)
        out2 = gc.eval_case(abi, oracle.lib, "gfo_", None, c2)
        ref = np.array([gc.psr_dopp_residual(q, c2["iono"], c2["pose"][q["lower_idx"], :3], c2["speed_bias"][q["lower_idx"], :3],
                                             c2["pose"][q["lower_idx"] + 1, :3], c2["speed_bias"][q["lower_idx"] + 1, :3],
                                             c2["rcv_dt"][q["frame"], q["sys_idx"]], c2["rcv_ddt"][q["frame"]], c2["yaw"], c2["anc"])[0] for q in c2["obs"]])
        assert np.abs(out2["r"] - ref).max() < 2e-6


def test_satellite_below_the_horizon_has_no_atmosphere_and_a_positive_weight(oracle):
    c = gc.gnss_case(8, n_per_frame=1)
    o = c["obs"][0]
    e, n, u = gc.enu_axes(22.3, 114.17)
    o["sv_pos"] = c["anc"] + 2.0e7 * (np.cos(-0.2) * n + np.sin(-0.2) * u)            # elevation -0.2 rad
    c["obs"] = [o]
    a = gc.eval_case(abi, oracle.lib, "gfo_", None, c)
    b = gc.eval_case(abi, oracle.lib, "gfo_", None, c, iono=None)
    assert np.array_equal(a["r"], b["r"]) and np.isfinite(a["r"]).all() and a["J"][0, 0, 12] > 0
