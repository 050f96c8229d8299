"""Synthetic GNSS windows and an independent numpy statement of the three GNSS factors' residuals (test infrastructure).

The numpy side is written from the published models rather than from the C restatements: geodetic coordinates by fixed-point
iteration on the latitude (not the closed form of gnss_utility.cpp:347), east / north / up unit vectors instead of a rotation
matrix product, Saastamoinen + Niell and Klobuchar as their formulas read (gnss_utility.cpp:774-899 holds the same constants).
`freeze` holds the terms the reference's analytic Jacobian leaves out (elevation weights, atmosphere, Sagnac) at their nominal
values so that central differences of it can be compared with that Jacobian tightly."""
import numpy as np

C_LIGHT, OMEGA_E, E2, A_EARTH = 2.99792458e8, 7.2921151467e-5, 6.69437999014e-3, 6378137.0
WINDOW = 10


def geo2ecef(lat_deg, lon_deg, h):
    lat, lon = np.radians(lat_deg), np.radians(lon_deg)
    N = A_EARTH / np.sqrt(1 - E2 * np.sin(lat) ** 2)
    return np.array([(N + h) * np.cos(lat) * np.cos(lon), (N + h) * np.cos(lat) * np.sin(lon), (N * (1 - E2) + h) * np.sin(lat)])


def ecef2geo_iter(x):
    p = np.hypot(x[0], x[1])
    lat = np.arctan2(x[2], p * (1 - E2))
    for _ in range(50):
        N = A_EARTH / np.sqrt(1 - E2 * np.sin(lat) ** 2)
        h = p / np.cos(lat) - N
        lat = np.arctan2(x[2], p * (1 - E2 * N / (N + h)))
    N = A_EARTH / np.sqrt(1 - E2 * np.sin(lat) ** 2)
    return np.array([np.degrees(lat), np.degrees(np.arctan2(x[1], x[0])), p / np.cos(lat) - N])


def enu_axes(lat_deg, lon_deg):
    lat, lon = np.radians(lat_deg), np.radians(lon_deg)
    east = np.array([-np.sin(lon), np.cos(lon), 0.0])
    north = np.array([-np.sin(lat) * np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat)])
    up = np.array([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)])
    return east, north, up


def azel(rcv, sat):
    lla = ecef2geo_iter(rcv)
    e, n, u = enu_axes(lla[0], lla[1])
    d = (sat - rcv) / np.linalg.norm(sat - rcv)
    az = np.arctan2(d @ e, d @ n)
    return (az + 2 * np.pi if az < 0 else az), np.arcsin(d @ u), lla


NMF = np.array([
    [1.2769934e-3, 1.2683230e-3, 1.2465397e-3, 1.2196049e-3, 1.2045996e-3], [2.9153695e-3, 2.9152299e-3, 2.9288445e-3, 2.9022565e-3, 2.9024912e-3],
    [62.610505e-3, 62.837393e-3, 63.721774e-3, 63.824265e-3, 64.258455e-3], [0.0, 1.2709626e-5, 2.6523662e-5, 3.4000452e-5, 4.1202191e-5],
    [0.0, 2.1414979e-5, 3.0160779e-5, 7.2562722e-5, 11.723375e-5], [0.0, 9.0128400e-5, 4.3497037e-5, 84.795348e-5, 170.37206e-5],
    [5.8021897e-4, 5.6794847e-4, 5.8118019e-4, 5.9727542e-4, 6.1641693e-4], [1.4275268e-3, 1.5138625e-3, 1.4572752e-3, 1.5007428e-3, 1.7599082e-3],
    [4.3472961e-2, 4.6729510e-2, 4.3908931e-2, 4.4626982e-2, 5.4736038e-2]])


def _interp(row, lat):      # table nodes at 15, 30, 45, 60, 75 degrees, clamped outside
    return float(np.interp(lat, [15.0, 30.0, 45.0, 60.0, 75.0], row))


def _herring(el, a, b, c):
    s = np.sin(el)
    return (1 + a / (1 + b / (1 + c))) / (s + a / (s + b / (s + c)))


def trop(doy, lla, el):
    lat, h = lla[0], lla[2]
    if h < -100 or h > 1e4 or el <= 0:
        return 0.0
    hh = max(h, 0.0)
    pres = 1013.25 * (1 - 2.2557e-5 * hh) ** 5.2568
    temp = 15.0 - 6.5e-3 * hh + 273.16
    e = 6.108 * 0.7 * np.exp((17.15 * temp - 4684.0) / (temp - 38.45))
    zhd = 0.0022768 * pres / (1 - 0.00266 * np.cos(2 * np.radians(lat)) - 0.00028 * hh / 1e3)
    zwd = 0.002277 * (1255.0 / temp + 0.05) * e
    cosy = np.cos(2 * np.pi * ((doy - 28.0) / 365.25 + (0.5 if lat < 0 else 0.0)))
    al = abs(lat)
    ah = [_interp(NMF[i], al) - _interp(NMF[i + 3], al) * cosy for i in range(3)]
    aw = [_interp(NMF[i + 6], al) for i in range(3)]
    dm = (1 / np.sin(el) - _herring(el, 2.53e-5, 5.49e-3, 1.14e-3)) * h / 1e3
    return (_herring(el, *ah) + dm) * zhd + _herring(el, *aw) * zwd


def klobuchar(tow, ion, lla, az, el):
    if ion is None or lla[2] < -1e3 or el <= 0:
        return 0.0
    psi = 0.0137 / (el / np.pi + 0.11) - 0.022
    phi = float(np.clip(lla[0] / 180 + psi * np.cos(az), -0.416, 0.416))
    lam = lla[1] / 180 + psi * np.sin(az) / np.cos(phi * np.pi)
    phi += 0.064 * np.cos((lam - 1.617) * np.pi)
    tt = (43200 * lam + tow) % 86400.0
    f = 1 + 16 * (0.53 - el / np.pi) ** 3
    amp = max(np.polyval(ion[3::-1], phi), 0.0)
    per = max(np.polyval(ion[7:3:-1], phi), 72000.0)
    x = 2 * np.pi * (tt - 50400) / per
    return C_LIGHT * f * (5e-9 + amp * (1 - x * x / 2 + x ** 4 / 24) if abs(x) < 1.57 else 5e-9)


def psr_dopp_residual(o, ion, Pi, Vi, Pj, Vj, rcv_dt, rcv_ddt, yaw, anc, freeze=None):
    """Returns (r[2], nominal) — `nominal` = the terms `freeze` would hold fixed, evaluated here."""
    lp, lv = o["ratio"] * Pi + (1 - o["ratio"]) * Pj, o["ratio"] * Vi + (1 - o["ratio"]) * Vj
    lla0 = ecef2geo_iter(anc)
    e, n, u = enu_axes(lla0[0], lla0[1])
    cy, sy = np.cos(yaw), np.sin(yaw)
    to_ecef = lambda v: (cy * v[0] - sy * v[1]) * e + (sy * v[0] + cy * v[1]) * n + v[2] * u
    P, V = to_ecef(lp) + anc, to_ecef(lv)
    sp, sv = np.asarray(o["sv_pos"]), np.asarray(o["sv_vel"])
    if freeze is None:
        az, el, lla = azel(P, sp)
        nominal = dict(sin2=np.sin(el) ** 2, atm=trop(o["doy"], lla, el) + klobuchar(o["tow"], ion, lla, az, el),
                       sag_p=OMEGA_E * (sp[0] * P[1] - sp[1] * P[0]) / C_LIGHT,
                       sag_d=OMEGA_E / C_LIGHT * (sv[0] * P[1] + sp[0] * V[1] - sv[1] * P[0] - sp[1] * V[0]))
    else:
        nominal = freeze
    d = sp - P
    rho = np.linalg.norm(d)
    wp, wd = nominal["sin2"] / o["pr_uura"] * 10.0, nominal["sin2"] / o["dp_uura"] * 50.0
    psr = rho + nominal["sag_p"] + rcv_dt - o["svdt"] * C_LIGHT + nominal["atm"] + o["tgd"] * C_LIGHT
    dop = (sv - V) @ d / rho + nominal["sag_d"] + rcv_ddt - o["svddt"] * C_LIGHT
    return np.array([(psr - o["psr"]) * wp, (dop + o["dopp"] * o["wavelength"]) * wd]), nominal


IONO = np.array([0.1118e-7, 0.2235e-7, -0.5960e-7, -0.1192e-6, 0.9626e5, 0.1311e6, -0.6554e5, -0.5898e6])   # a broadcast-like set


def gnss_case(seed, n_per_frame=6, lat=22.3, lon=114.17, h=30.0, el_min=8.0):
    rng = np.random.default_rng(seed)
    anc = geo2ecef(lat, lon, h)
    pose = np.zeros((WINDOW + 1, 7))
    pose[:, :3] = np.cumsum(rng.normal(0, 1.5, (WINDOW + 1, 3)), axis=0) + rng.normal(0, 20, 3)
    pose[:, 2] *= 0.1
    pose[:, 6] = 1.0
    sb = np.zeros((WINDOW + 1, 9))
    sb[:, :3] = rng.normal(0, 3, (WINDOW + 1, 3))
    yaw = rng.uniform(-np.pi, np.pi)
    rcv_ddt = 30.0 + np.cumsum(rng.normal(0, 0.05, WINDOW + 1))
    frame_dt = rng.uniform(0.08, 0.12, WINDOW)
    rcv_dt = np.zeros((WINDOW + 1, 4))
    rcv_dt[0] = rng.uniform(-2e5, 2e5, 4)
    for i in range(WINDOW):
        rcv_dt[i + 1] = rcv_dt[i] + 0.5 * (rcv_ddt[i] + rcv_ddt[i + 1]) * frame_dt[i] + rng.normal(0, 0.02, 4)
    e, n, u = enu_axes(lat, lon)
    obs = []
    for i in range(WINDOW + 1):
        for _ in range(n_per_frame):
            lower = int(rng.integers(max(i - 1, 0), min(i, WINDOW - 1) + 1))
            az, el = rng.uniform(0, 2 * np.pi), np.radians(rng.uniform(el_min, 88.0))
            los = np.cos(el) * np.sin(az) * e + np.cos(el) * np.cos(az) * n + np.sin(el) * u
            o = dict(sv_pos=anc + rng.uniform(2.0e7, 2.55e7) * los, sv_vel=rng.normal(0, 1800.0, 3), svdt=rng.uniform(-5e-4, 5e-4),
                     svddt=rng.normal(0, 1e-11), tgd=rng.normal(0, 6e-9), pr_uura=rng.uniform(2.0, 6.0), dp_uura=rng.uniform(0.2, 0.6),
                     wavelength=C_LIGHT / rng.choice([1575.42e6, 1602.0e6, 1561.098e6]), ratio=float(rng.uniform(0, 1)),
                     doy=rng.uniform(1, 366), tow=rng.uniform(0, 604800), frame=i, lower_idx=lower, sys_idx=int(rng.integers(0, 4)), psr=0.0, dopp=0.0)
            r, _ = psr_dopp_residual(o, IONO, pose[lower, :3], sb[lower, :3], pose[lower + 1, :3], sb[lower + 1, :3], rcv_dt[i, o["sys_idx"]],
                                     rcv_ddt[i], yaw, anc)
            # r = (estimate - 0) * weight with psr = dopp = 0: back out the measurement that leaves metre / decimetre-per-second noise
            _, nom = psr_dopp_residual(o, IONO, pose[lower, :3], sb[lower, :3], pose[lower + 1, :3], sb[lower + 1, :3], rcv_dt[i, o["sys_idx"]],
                                       rcv_ddt[i], yaw, anc)
            wp, wd = nom["sin2"] / o["pr_uura"] * 10.0, nom["sin2"] / o["dp_uura"] * 50.0
            o["psr"] = r[0] / wp + rng.normal(0, 2.0)
            o["dopp"] = -(r[1] / wd + rng.normal(0, 0.2)) / o["wavelength"]
            obs.append(o)
    return dict(obs=obs, iono=IONO, pose=pose, speed_bias=sb, rcv_dt=rcv_dt, rcv_ddt=rcv_ddt, yaw=yaw, anc=anc, frame_dt=frame_dt, ddt_weight=10.0)


def eval_case(abi, lib, prefix, ctx, c, iono="case", want_J=True):
    return abi.gnss_eval(lib, prefix, ctx, c["obs"], c["iono"] if isinstance(iono, str) else iono, c["pose"], c["speed_bias"], c["rcv_dt"], c["rcv_ddt"],
                         c["yaw"], c["anc"], c["frame_dt"], c["ddt_weight"], want_J=want_J)
