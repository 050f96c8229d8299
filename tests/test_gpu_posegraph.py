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


@pytest.mark.parametrize("seed,n,fix_every,max_it,kick", [(11, 120, 4, 15, 0.0), (12, 90, 3, 15, 0.0), (13, 150, 5, 0, 0.0), (14, 150, 5, 1, 0.0),
                                                           (15, 200, 6, 12, 2.0), (16, 64, 2, 15, 8.0), (17, 333, 9, 15, 0.5), (18, 40, 40, 15, 0.0)])
def test_the_loop_on_the_device_takes_the_oracles_branches(be, oracle, seed, n, fix_every, max_it, kick):
    """The Levenberg-Marquardt loop runs on the device since round 6 (PgState in csrc/gfbe_posegraph.hip): iteration caps of 0 and 1, runs that
    end on a tolerance before the cap (every later pass of the enqueued launches must do nothing), a graph with one fix only, and starting
    points thrown far off (`kick` metres / tenths of a radian: rejected and invalid steps) — each with the oracle's iteration count, accept /
    reject sequence, termination and costs."""
    g = synth.pose_graph(n=n, seed=seed, fix_every=fix_every)
    if kick > 0.0:
        rng = np.random.default_rng(seed)
        pose = g["pose"].copy()
        pose[:, :3] += rng.normal(0, kick, (n, 3))
        q = pose[:, 3:] + rng.normal(0, 0.1 * kick, (n, 4))
        pose[:, 3:] = q / np.linalg.norm(q, axis=1)[:, None]
        g["pose"] = pose
    ref = abi.PoseGraph(oracle.lib, "gfo_", None).solve(g, max_iterations=max_it)
    got = abi.PoseGraph(be.lib, "gfbe_", be.ctx).solve(g, max_iterations=max_it)
    sr, sg = ref["summary"], got["summary"]
    assert (sg["iterations"], sg["accepted"], sg["termination"], sg["status"]) == (sr["iterations"], sr["accepted"], sr["termination"], sr["status"])
    floor = 1e-12 * sr["initial_cost"]      # (a graph that is solved exactly ends on costs of 1e-15 .. 1e-20: rounding, not digits)
    np.testing.assert_allclose(sg["cost_history"], sr["cost_history"], rtol=1e-8, atol=floor)
    assert abs(sg["final_cost"] - sr["final_cost"]) <= 1e-8 * sr["final_cost"] + floor
    assert np.abs(got["pose"][:, :3] - ref["pose"][:, :3]).max() < 1e-7
    assert np.abs(got["pose"][:, 3:] - ref["pose"][:, 3:]).max() < 1e-8


def test_bad_graph_is_rejected(be):
    g = synth.pose_graph(n=10)
    g["rel_i"] = g["rel_i"].copy()
    g["rel_i"][3] = 9          # (9, 10): out of range
    with pytest.raises(RuntimeError, match="status 3"):
        abi.PoseGraph(be.lib, "gfbe_", be.ctx).solve(g)
