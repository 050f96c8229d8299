"""The product's factor arithmetic (csrc/gfbe_factors.h, __host__ __device__) compiled for the
host by tests/host_shim.cpp and pinned against the CPU oracle — runs without a GPU. The same
functions run inside the HIP kernels; tests/test_gpu_*.py repeat the comparison through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "_build", "libhost_shim.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def shim():
    return build_shim()


def build_shim():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "tests", "host_shim.cpp")
    deps = [src, os.path.join(ROOT, "ground-fusion2_amd", "csrc", "gfbe_factors.h"),
            os.path.join(ROOT, "ground-fusion2_amd", "csrc", "gfbe_math.h"), os.path.join(ROOT, "ground-fusion2_amd", "csrc", "gfbe_gnss.h"),
            os.path.join(ROOT, "include", "gfbe.h")]
    if not os.path.exists(SHIM) or any(os.path.getmtime(d) > os.path.getmtime(SHIM) for d in deps):
        os.makedirs(os.path.dirname(SHIM), exist_ok=True)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                        "-o", SHIM, src], check=True)
    return C.CDLL(SHIM)


def shim_eval(shim, snap, robust):
    wh = abi.WindowHolder(snap)
    K, ni, nw = wh.n_vis, wh.c.n_imu, wh.c.n_wheel
    out = dict(vis_r=np.zeros((K, 2)), vis_J=np.zeros((K, 2, 20)), imu_r=np.zeros((ni, 15)),
               imu_J=np.zeros((ni, 15, 30)), wheel_r=np.zeros((nw, 6)), wheel_J=np.zeros((nw, 6, 22)))
    opt = abi.default_options()
    rc = shim.shim_eval_factors(C.byref(opt), C.byref(wh.c), int(robust), abi._pd(out["vis_r"]), abi._pd(out["vis_J"]),
                                abi._pd(out["imu_r"]), abi._pd(out["imu_J"]), abi._pd(out["wheel_r"]),
                                abi._pd(out["wheel_J"]), None)
    assert rc == 0
    return out


@pytest.mark.parametrize("robust", [False, True])
def test_device_factor_math_matches_oracle(shim, oracle, robust):
    scn = synth.Scenario(seed=41, n_landmarks=120, use_wheel=True)
    snap = scn.window(0)
    snap["ix_wheel"] = np.array([1.01, 0.98, 1.02])
    snap["td"], snap["td_wheel"] = 0.003, -0.004
    want = oracle.eval_factors(snap, robustify=robust)
    got = shim_eval(shim, snap, robust)
    for k in ("vis_r", "vis_J", "wheel_r", "wheel_J"):
        sc = max(1.0, np.abs(want[k]).max())
        assert np.abs(got[k] - want[k]).max() < 1e-12 * sc, k
    # IMU: sqrt_info of a cov with condition ~1e12 -> same algorithm, same roundoff class
    for k in ("imu_r", "imu_J"):
        sc = max(1.0, np.abs(want[k]).max())
        assert np.abs(got[k] - want[k]).max() < 1e-9 * sc, k


def test_device_gnss_math_matches_oracle(shim, oracle):
    """csrc/gfbe_gnss.h on the host against oracle/gfo_gnss.cpp: pseudo-ranges of 2.5e7 m in doubles, weights up to 250."""
    import gnss_cases as gc

    class Shim:      # abi.gnss_eval drives any library exporting <prefix>gnss_eval with the C ABI's argument list
        def __init__(self, lib):
            self.lib = lib

        def __getattr__(self, name):
            assert name == "shim_gnss_eval"
            f = self.lib.shim_gnss_eval

            def call(ctx, n, arr, io, st, g, fdt, w, r, J, rc, rs, cost):
                return f(n, arr, io, st, g, fdt, C.c_double(w), r, J, rc, rs)
            return _Holder(call)

    class _Holder:
        def __init__(self, fn):
            self.fn, self.restype, self.argtypes = fn, None, None

        def __call__(self, *a):
            return self.fn(*a)

    for seed, lat in [(11, 22.3), (12, -50.0), (13, 80.0)]:
        c = gc.gnss_case(seed, lat=lat)
        want = gc.eval_case(abi, oracle.lib, "gfo_", None, c)
        got = gc.eval_case(abi, Shim(shim), "shim_", None, c)
        assert np.abs(got["r"] - want["r"]).max() < 1e-5               # 4e-9 m resolution x weight 250 x a few operations
        assert np.abs(got["J"] - want["J"]).max() < 1e-9 * np.abs(want["J"]).max()
        np.testing.assert_allclose(got["r_dt_ddt"], want["r_dt_ddt"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(got["r_smooth"], want["r_smooth"], rtol=0, atol=1e-12)
        noi = gc.eval_case(abi, Shim(shim), "shim_", None, c, iono=None)
        assert np.abs(noi["r"] - gc.eval_case(abi, oracle.lib, "gfo_", None, c, iono=None)["r"]).max() < 1e-5
