"""examples/stream_loop.cpp: the one-robot frame loop (feature tables -> triangulate -> optimization() -> outlier check -> slide)
driven by a COMPILED caller through include/gfbe.h alone. Without a GPU it must fail loudly at gfbe_create; on an MI355X its
trajectory equals the one the Python driver (ground-fusion2_amd/stream.py::run_stream, device hand-over) produces from the same
stream: both make the same library calls, only the dead reckoning between two frames is a second implementation (C++ / numpy)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from _gfbe_import import gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    gf.build_native()
    exe = tmp_path / "stream_loop"
    libdir = os.path.dirname(gf.lib_path())
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "stream_loop.cpp"), "-L", libdir, "-lgfbe", "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    return exe


def dump(tmp_path, S):
    import importlib.util
    spec = importlib.util.spec_from_file_location("dump_stream", os.path.join(ROOT, "tools", "dump_stream.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = tmp_path / "stream.bin"
    mod.dump(str(path), S)
    return path


def test_stream_loop_compiles_reads_its_stream_and_fails_loudly_without_a_gpu(tmp_path):
    exe = build(tmp_path)
    S = gf.stream.Stream(seed=5, n_kf=14, new_per_frame=40)
    path = dump(tmp_path, S)
    out = subprocess.run([str(exe), str(path), str(tmp_path / "traj.bin")], capture_output=True, text=True, timeout=120)
    if out.returncode != 0:
        assert "gfbe_create" in out.stderr and "no CPU fallback" in out.stderr, out.stdout + out.stderr
    else:
        assert "compiled loop" in out.stdout
    bad = subprocess.run([str(exe), str(exe), str(tmp_path / "traj.bin")], capture_output=True, text=True, timeout=120)     # (not a stream file)
    assert bad.returncode == 2 and "cannot read" in bad.stderr


@pytest.mark.gpu
def test_stream_loop_reproduces_the_python_driven_loop(tmp_path):
    abi, stream = gf.abi, gf.stream
    exe = build(tmp_path)
    S = stream.Stream(seed=3, n_kf=28, new_per_frame=50)
    path, traj_path = dump(tmp_path, S), tmp_path / "traj.bin"
    out = subprocess.run([str(exe), str(path), str(traj_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(traj_path, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    traj = np.frombuffer(raw[4:4 + 56 * n], np.float64).reshape(n, 7)
    costs = np.frombuffer(raw[4 + 56 * n:4 + 64 * n], np.float64)
    iters, flags, n_lm, n_out = np.frombuffer(raw[4 + 64 * n:], np.int32).reshape(4, n)
    be = gf.Backend(device=0)
    T = abi.FeatureTables(be.lib, "gfbe_", be.ctx, 1, 16384, options=dict(min_parallax=14.0 / 600, depth_threshold=6.0))
    outliers, check = [], T.check_outliers

    def counted(*a, **k):
        r = check(*a, **k)
        outliers.append(len(r[0]))
        return r
    T.check_outliers = counted
    slide = be.lib.gfbe_slide_window_state
    ref = stream.run_stream(be, T, S, lambda st, flag: slide(C.byref(st), int(flag)), device_handoff=True)
    T.close()
    assert n == len(ref["traj"]) == S.n_kf - abi.WINDOW_SIZE
    # every discrete outcome: keyframe decisions, iteration counts, landmarks in the window, features the consistency check removes
    assert flags.tolist() == ref["flags"] and iters.tolist() == ref["iterations"]
    assert n_lm.tolist() == ref["n_landmarks"] and n_out.tolist() == outliers
    assert len(set(flags.tolist())) == 2                                      # (both marginalisation flavours occur)
    # the first solve has identical inputs; from the second on the two dead reckonings (numpy / C++) differ in the last bit of the
    # newest pose (3e-14 m), which solves that stop on the 8-iteration budget amplify: measured 3e-10 m and 6e-6 of the final cost
    assert np.array_equal(traj[0], ref["traj"][0]) and costs[0] == ref["final_cost"][0]
    assert np.abs(traj - ref["traj"]).max() < 1e-8
    np.testing.assert_allclose(costs, ref["final_cost"], rtol=1e-4)
