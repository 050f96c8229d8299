"""Windows with GNSS inside the solve (gfbe_window.gnss_ready; estimator.cpp:2965-3002, 3239-3291, 3462-3496): shared by the CPU
and GPU tests. The measurements come from the scenario's true trajectory through the independent numpy statement of the
pseudo-range / Doppler model in tests/gnss_cases.py (not through the C restatements), so that a solve has something to converge to."""
import numpy as np

from _gfbe_import import gf
import gnss_cases as gc

abi, synth = gf.abi, gf.synth
W = abi.WINDOW_SIZE


class GnssTruth:
    """Receiver clock, anchor and yaw of a whole scenario run + the observations of every keyframe (deterministic in the seed)."""

    def __init__(self, scn, seed, n_per_frame=8, lat=22.3, lon=114.17, h=30.0, two_sided=True):
        self.scn = scn
        rng = np.random.default_rng(seed + 4242)
        self.anc = gc.geo2ecef(lat, lon, h)
        self.yaw = float(rng.uniform(-3.0, 3.0))
        n = scn.n_kf
        self.frame_dt = np.diff(scn.kf_t)
        self.ddt = 30.0 + np.cumsum(rng.normal(0, 0.05, n))
        self.dt = np.zeros((n, 4))
        self.dt[0] = rng.uniform(-2e5, 2e5, 4)
        for i in range(n - 1):
            self.dt[i + 1] = self.dt[i] + 0.5 * (self.ddt[i] + self.ddt[i + 1]) * self.frame_dt[i]
        e, nn, u = gc.enu_axes(lat, lon)
        self.epochs = []          # per keyframe: list of (offset of the observation time from the keyframe, observation dict)
        for g in range(n):
            obs = []
            off0 = float(rng.uniform(-0.02, 0.02))
            for q in range(n_per_frame):
                off = -off0 if (two_sided and q == 0) else off0        # one satellite on the other side of the keyframe time
                t = scn.kf_t[g] + off
                p, _, v, _, _ = scn.kinematics(t)
                az, el = rng.uniform(0, 2 * np.pi), np.radians(rng.uniform(12.0, 88.0))
                los = np.cos(el) * np.sin(az) * e + np.cos(el) * np.cos(az) * nn + np.sin(el) * u
                o = dict(sv_pos=self.anc + rng.uniform(2.0e7, 2.55e7) * los, sv_vel=rng.normal(0, 1800.0, 3), svdt=rng.uniform(-5e-4, 5e-4),
                         svddt=rng.normal(0, 1e-11), tgd=rng.normal(0, 6e-9), pr_uura=rng.uniform(2.0, 6.0), dp_uura=rng.uniform(0.2, 0.6),
                         wavelength=gc.C_LIGHT / rng.choice([1575.42e6, 1602.0e6, 1561.098e6]), ratio=1.0, doy=rng.uniform(1, 366),
                         tow=rng.uniform(0, 604800), frame=0, lower_idx=0, sys_idx=int(rng.integers(0, 4)), psr=0.0, dopp=0.0)
                r, nom = gc.psr_dopp_residual(o, gc.IONO, p, v, p, v, self.dt[g, o["sys_idx"]], self.ddt[g], self.yaw, self.anc)
                wp, wd = nom["sin2"] / o["pr_uura"] * 10.0, nom["sin2"] / o["dp_uura"] * 50.0
                o["psr"] = r[0] / wp + rng.normal(0, 1.0)
                o["dopp"] = -(r[1] / wd + rng.normal(0, 0.1)) / o["wavelength"]
                obs.append((off, o))
            self.epochs.append(obs)

    def window_obs(self, k0):
        """Observations of window k0 the way estimator.cpp:3245-3263 indexes them: frame i, lower_idx, ts_ratio."""
        out = []
        t = self.scn.kf_t
        for i in range(W + 1):
            for off, o in self.epochs[k0 + i]:
                ts = t[k0 + i] + off
                lower = (0 if i == 0 else i - 1) if t[k0 + i] > ts else (W - 1 if i == W else i)
                ratio = (t[k0 + lower + 1] - ts) / (t[k0 + lower + 1] - t[k0 + lower])
                out.append(dict(o, frame=i, lower_idx=lower, ratio=float(ratio)))
        return out

    def state(self, k0, seed, noise=True):
        rng = np.random.default_rng(seed + 99 + k0)
        sc = 1.0 if noise else 0.0
        return dict(rcv_dt=self.dt[k0:k0 + W + 1] + sc * rng.normal(0, 0.5, (W + 1, 4)), rcv_ddt=self.ddt[k0:k0 + W + 1] + sc * rng.normal(0, 0.02, W + 1),
                    yaw_enu_local=self.yaw + sc * 0.002, anc_ecef=self.anc + sc * rng.normal(0, 1.0, 3))

    def block(self, k0, ready=1):
        return dict(obs=self.window_obs(k0), iono=gc.IONO, frame_dt=self.frame_dt[k0:k0 + W], ddt_weight=10.0, ready=ready)


def gnss_window(seed=81, L=150, n_per_frame=8, noise=True, anchor=False, **kw):
    """First window of a scenario with GNSS ready: returns (scenario, GNSS truth, snapshot)."""
    scn = synth.Scenario(seed=seed, n_landmarks=L, use_wheel=True, noise=noise)
    tru = GnssTruth(scn, seed, n_per_frame=n_per_frame, **kw)
    snap = scn.window(0)
    snap["gnss"], snap["gnss_state"] = tru.block(0), tru.state(0, seed, noise)
    if anchor:      # first_optimization && GNSS_ENABLE (estimator.cpp:3004-3012)
        snap["anchor"] = dict(pose=snap["pose"][0].copy(), sqrt_info=120.0)
    return scn, tru, snap


def next_gnss_window(scn, tru, res, seed=81):
    """The following window: slideWindow's shift of the state (estimator.cpp:3736-3743 for the receiver clock), the prior of `res`."""
    st = synth.shift_state_for_next_window(scn, res["state"], 1)
    nxt = scn.window(1, state=st, prior=res["prior"])
    g, fresh = res["state"]["gnss_state"], tru.state(1, seed)
    nxt["gnss_state"] = dict(rcv_dt=np.vstack([g["rcv_dt"][1:], fresh["rcv_dt"][-1:]]), rcv_ddt=np.concatenate([g["rcv_ddt"][1:], fresh["rcv_ddt"][-1:]]),
                             yaw_enu_local=g["yaw_enu_local"], anc_ecef=g["anc_ecef"])
    nxt["gnss"] = tru.block(1)
    return nxt
