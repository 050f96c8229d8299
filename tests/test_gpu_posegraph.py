"""global_fusion pose graph on the device (csrc/gfbe_posegraph.hip) vs the CPU oracle, through the C ABI
(BASELINE configs[3]: 5 000 poses). Factor blocks 1e-12 relative (same formulas); the
Levenberg-Marquardt solve: identical accept / reject sequence, costs 1e-9 relative, poses 1e-8 m — the device solves the
block-tridiagonal system by parallel cyclic reduction, the oracle by a block Cholesky recurrence."""
import time

import numpy as np
import pytest

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


def test_factor_blocks_match_oracle(be, oracle):
    g = synth.pose_graph(n=300, seed=3, fix_every=7)
    a = abi.PoseGraph(oracle.lib, "gfo_", None).eval(g)
    b = abi.PoseGraph(be.lib, "gfbe_", be.ctx).eval(g)
    for k in ("rel_r", "rel_J", "fix_r"):
        assert np.abs(a[k] - b[k]).max() <= 1e-12 * max(1.0, np.abs(a[k]).max()), k
    assert abs(a["cost"] - b["cost"]) <= 1e-12 * a["cost"]


@pytest.mark.parametrize("n,fix_every", [(5000, 10), (257, 3), (2, 1)])
def test_solve_matches_oracle(be, oracle, n, fix_every):
    g = synth.pose_graph(n=n, fix_every=fix_every)
    ref = abi.PoseGraph(oracle.lib, "gfo_", None).solve(g, max_iterations=5)
    t0 = time.perf_counter()
    got = abi.PoseGraph(be.lib, "gfbe_", be.ctx).solve(g, max_iterations=5)
    sr, sg = ref["summary"], got["summary"]
    assert sg["iterations"] == sr["iterations"] and sg["accepted"] == sr["accepted"] and sg["termination"] == sr["termination"]
    np.testing.assert_allclose(sg["cost_history"], sr["cost_history"], rtol=1e-9)
    assert np.abs(got["pose"][:, :3] - ref["pose"][:, :3]).max() < 1e-8
    assert np.abs(got["pose"][:, 3:] - ref["pose"][:, 3:]).max() < 1e-9
    if n == 5000:
        err = np.linalg.norm(got["pose"][:, :3] - g["truth"][:, :3], axis=1).mean()
        assert err < 0.5


def test_bad_graph_is_rejected(be):
    g = synth.pose_graph(n=10)
    g["rel_i"] = g["rel_i"].copy()
    g["rel_i"][3] = 9          # (9, 10): out of range
    with pytest.raises(RuntimeError, match="status 3"):
        abi.PoseGraph(be.lib, "gfbe_", be.ctx).solve(g)
