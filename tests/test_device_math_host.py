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
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "tests", "host_shim.cpp")
    deps = [src, os.path.join(ROOT, "ground-fusion2_amd", "csrc", "gfbe_factors.h"),
            os.path.join(ROOT, "ground-fusion2_amd", "csrc", "gfbe_math.h")]
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
