"""The drop-in boundary from plain C: examples/gfbe_minimal.c includes only include/gfbe.h, is compiled as C99 with -Wall -Wextra
-Werror, linked against libgfbe.so and run. Without a GPU it must report GFBE_NO_DEVICE (there is no CPU fallback); on an MI355X
it solves its hand-made window."""
import os
import subprocess

import pytest

from _gfbe_import import gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path, cxx=False):
    gf.build_native()
    exe = tmp_path / ("estimator_binding" if cxx else "gfbe_minimal")
    libdir = os.path.dirname(gf.lib_path())
    cc = ["g++", "-std=c++17"] if cxx else ["gcc", "-std=c99"]
    src = "estimator_binding.cpp" if cxx else "gfbe_minimal.c"
    subprocess.run(cc + ["-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", src),
                         "-L", libdir, "-lgfbe", "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)], check=True)
    return exe


def test_c_caller_compiles_links_and_fails_loudly_without_a_gpu(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GFBE_NO_DEVICE" in out.stdout or "iterations" in out.stdout


@pytest.mark.gpu
def test_c_caller_solves_its_window(tmp_path):
    exe = build(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "iterations" in out.stdout and "gfbe 0.1.0" in out.stdout


def test_estimator_binding_compiles_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/estimator_binding.cpp: the INTEGRATION.md patch as compiling C++ over stand-alone copies of the Estimator members."""
    exe = build(tmp_path, cxx=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GFBE_NO_DEVICE" in out.stdout or "optimization() #2" in out.stdout


@pytest.mark.gpu
def test_estimator_binding_runs_two_frames_with_a_carried_prior(tmp_path):
    exe = build(tmp_path, cxx=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("optimization()")]
    assert len(lines) == 2 and all("prior valid" in ln for ln in lines)
