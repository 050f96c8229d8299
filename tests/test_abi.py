"""CPU-side checks of the drop-in boundary: the HIP library loads without a GPU, exports every
symbol include/gfbe.h declares, the integer landmark bookkeeping (a13) is bit-exact against the
oracle, and compute entry points fail loudly (no CPU fallback) when no device is given."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    gf.build_native()
    return C.CDLL(gf.lib_path())


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "gfbe.h")).read()
    declared = sorted(set(re.findall(r"\b(gfbe_[a-z_0-9]+)\s*\(", hdr)))
    declared = [d for d in declared if d not in ("gfbe_allreduce_fn",)]
    assert set(declared) == set(gf.backend.EXPORTS), set(declared) ^ set(gf.backend.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header(lib):
    assert C.sizeof(abi.State) == 259 * 8
    assert C.sizeof(abi.GnssObs) == 19 * 8 and C.sizeof(abi.Window) == 2536
    assert C.sizeof(abi.ImuPreint) == 467 * 8
    assert C.sizeof(abi.WheelPreint) == 78 * 8
    o = abi.Options()
    lib.gfbe_default_options(C.byref(o))
    d = abi.default_options()
    for f, _ in abi.Options._fields_:
        assert getattr(o, f) == getattr(d, f), f


def test_options_struct_size_is_checked(lib):
    """gfbe_options.struct_size: gfbe_default_options writes the library's sizeof, gfbe_options_size returns it, and
    gfbe_create refuses a struct of another size instead of reading past it."""
    lib.gfbe_options_size.restype = abi.c_i
    assert lib.gfbe_options_size() == C.sizeof(abi.Options)
    o = abi.Options()
    lib.gfbe_default_options(C.byref(o))
    assert o.struct_size == C.sizeof(abi.Options) == abi.default_options().struct_size
    lib.gfbe_create.restype = abi.c_i
    lib.gfbe_last_error.restype = C.c_char_p
    for size, want in ((o.struct_size, abi.OK), (o.struct_size - 4, abi.BAD_INPUT), (0, abi.BAD_INPUT)):
        o.struct_size = size
        ctx = C.c_void_p()
        assert lib.gfbe_create(C.byref(ctx), -1, C.byref(o)) == want
        if want != abi.OK:
            assert b"ABI mismatch" in lib.gfbe_last_error(ctx)
        lib.gfbe_destroy(ctx)


class HostOnly(abi.CApi):
    prefix = "gfbe_"

    def __init__(self, lib):
        self.lib = abi.bind(lib, "gfbe_")
        self.ctx = C.c_void_p()
        lib.gfbe_create.restype = abi.c_i
        assert lib.gfbe_create(C.byref(self.ctx), -1, None) == abi.OK
        self.head = self.ctx


def test_bookkeeping_bit_exact_vs_oracle(lib, oracle):
    host = HostOnly(lib)
    for seed, extra in ((1, 0), (2, 30), (3, 100)):
        scn = synth.Scenario(seed=seed, n_landmarks=80)
        fl = scn.feature_list(0, extra_short=extra)
        fl["estimate_flag"][::5] = 1
        for only0 in (False, True):
            a, b = host.build_visual_factors(fl, only0), oracle.build_visual_factors(fl, only0)
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        lam = a["para_feature"].copy()
        lam[::9] *= -1
        ea, fa = host.set_depth(fl, lam)
        eb, fb = oracle.set_depth(fl, lam)
        np.testing.assert_array_equal(ea, eb)
        np.testing.assert_array_equal(fa, fb)


def test_empty_and_ragged_feature_lists(lib, oracle):
    host = HostOnly(lib)
    empty = dict(start_frame=np.zeros(0, np.int32), n_obs=np.zeros(0, np.int32), obs=np.zeros((0, 7)),
                 obs_td=np.zeros(0), estimated_depth=np.zeros(0), estimate_flag=np.zeros(0, np.int32))
    out = host.build_visual_factors(empty)
    assert len(out["vis_imu_i"]) == 0 and len(out["para_feature"]) == 0
    # only short tracks: nothing survives the used_num >= 4 filter (feature_manager.cpp:49)
    short = dict(start_frame=np.array([0, 3, 8], np.int32), n_obs=np.array([3, 1, 2], np.int32),
                 obs=np.random.default_rng(0).normal(size=(6, 7)), obs_td=np.zeros(6),
                 estimated_depth=np.ones(3), estimate_flag=np.zeros(3, np.int32))
    out = host.build_visual_factors(short)
    assert len(out["vis_imu_i"]) == 0 and len(out["para_feature"]) == 0
    # maximum track: 11 observations from frame 0 -> 10 factors, imu_j = 1..10
    full = dict(start_frame=np.array([0], np.int32), n_obs=np.array([11], np.int32),
                obs=np.random.default_rng(1).normal(size=(11, 7)), obs_td=np.arange(11.0),
                estimated_depth=np.array([2.0]), estimate_flag=np.array([1], np.int32))
    a, b = host.build_visual_factors(full), oracle.build_visual_factors(full)
    assert a["vis_imu_j"].tolist() == list(range(1, 11)) and a["feature_const"].tolist() == [1]
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])


def test_compute_fails_loudly_without_device(lib):
    host = HostOnly(lib)
    scn = synth.Scenario(seed=4, n_landmarks=10)
    wh = abi.WindowHolder(scn.window(0))
    st, sm = abi.State(), abi.Summary()
    feat = np.zeros(wh.n_feature)
    lib.gfbe_solve_window.restype = abi.c_i
    rc = lib.gfbe_solve_window(host.ctx, C.byref(wh.c), abi.MARGIN_NONE, C.byref(st), abi._pd(feat), None, C.byref(sm))
    assert rc == abi.NO_DEVICE
    lib.gfbe_last_error.restype = C.c_char_p
    lib.gfbe_last_error.argtypes = [C.c_void_p]
    assert b"no CPU fallback" in lib.gfbe_last_error(host.ctx)


def test_backend_raises_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gf.BackendError):
        gf.Backend(device=0)


def test_every_public_struct_matches_the_c_header(tmp_path):
    """sizeof and every field offset of the ctypes mirrors (abi.py) against what a C compiler makes of include/gfbe.h."""
    import subprocess
    pairs = [("gfbe_state", abi.State), ("gfbe_imu_preint", abi.ImuPreint), ("gfbe_wheel_preint", abi.WheelPreint),
             ("gfbe_visual", abi.Visual), ("gfbe_prior", abi.Prior), ("gfbe_lio_block", abi.LioBlock), ("gfbe_window", abi.Window),
             ("gfbe_options", abi.Options), ("gfbe_summary", abi.Summary), ("gfbe_feature_list", abi.FeatureList),
             ("gfbe_ftab_options", abi.FtabOptions), ("gfbe_gnss_obs", abi.GnssObs), ("gfbe_gnss_state", abi.GnssState)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gfbe.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, f)]) == getattr(cls, f).offset, "%s.%s" % (cname, f)
