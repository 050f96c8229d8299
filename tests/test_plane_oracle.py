"""PlaneFactor / PoseAnchorFactor INSIDE the window solve and the marginalisation (SURVEY.md section 8 a15 / f2;
estimator.cpp:3120-3136, 3214-3228, 3441-3448, 3004-3012), CPU side: the oracle against the independent numpy trust-region loop
(tests/ceres_trust_region_np.py, which stacks the plane / anchor rows from the stand-alone evaluators), the block bookkeeping of
the prior (para_plane_R travels as a 4-wide block WITHOUT a manifold: marginalization_factor.cpp:140-143) and the Schur identity
of the marginalisation."""
import numpy as np
import pytest

from _gfbe_import import gf
import ceres_trust_region_np as ctr
from plane_cases import next_plane_window, plane_window

abi, synth = gf.abi, gf.synth


def test_plane_and_anchor_in_the_solve_match_the_independent_loop(oracle):
    scn, snap = plane_window()
    w = oracle.solve(snap, abi.MARGIN_OLD)
    r = ctr.solve(oracle, snap)
    assert w["summary"]["accepted"] == r["accepted"] and w["summary"]["termination"] == r["termination"]
    np.testing.assert_allclose(w["summary"]["cost_history"], r["cost_history"], rtol=1e-6)
    assert abs(w["summary"]["final_radius"] - r["final_radius"]) < 1e-6 * r["final_radius"]
    # the plane blocks are estimated (OrientationSubsetParameterization({2}): tangent component 2 held, estimator.cpp:3122)
    assert np.abs(w["state"]["plane_R"] - snap["plane_R"]).max() > 1e-6 and abs(w["state"]["plane_Z"] - snap["plane_Z"]) > 1e-4
    assert abs(np.linalg.norm(w["state"]["plane_R"]) - 1.0) < 1e-14
    # the second window carries the prior with the plane blocks
    snap2 = next_plane_window(scn, snap, w)
    w2, r2 = oracle.solve(snap2, abi.MARGIN_OLD), ctr.solve(oracle, snap2)
    assert w2["summary"]["accepted"] == r2["accepted"]
    np.testing.assert_allclose(w2["summary"]["cost_history"], r2["cost_history"], rtol=1e-4)   # (rejected steps: an unsettled run)


def test_plane_blocks_in_the_prior(oracle):
    scn, snap = plane_window(anchor=False)
    w = oracle.solve(snap, abi.MARGIN_OLD)
    pr = w["prior"]
    ids, sizes = pr["block_id"].tolist(), pr["block_size"].tolist()
    assert abi.BLK_PLANE_R in ids and abi.BLK_PLANE_Z in ids
    assert sizes[ids.index(abi.BLK_PLANE_R)] == 4 and sizes[ids.index(abi.BLK_PLANE_Z)] == 1     # 4-wide, no manifold (App. A.3)
    assert pr["n"] == sum(6 if s == 7 else s for s in sizes)
    # the quaternion's 4th column carries no information: the factor's 3 x 4 block has a zero last column (plane_factor.h:96-102)
    A = pr["J0"].T @ pr["J0"]
    i4 = int(pr["block_idx"][ids.index(abi.BLK_PLANE_R)]) + 3
    assert np.abs(A[i4]).max() < 1e-7 * np.abs(A).max()
    # marginalisation identity against a numpy Schur complement of the oracle's own A, b (gfo_marginalize)
    snap_at = dict(snap)
    snap_at.update({k: w["state"][k] for k in ("pose", "speed_bias", "ex_pose", "ex_pose_wheel", "ix_wheel", "td", "td_wheel", "plane_R", "plane_Z")})
    snap_at["para_feature"] = w["feature"]
    pr2, Ap, bp, rc = oracle.marginalize(snap_at, abi.MARGIN_OLD)
    assert rc == 0 and pr2["block_id"].tolist() == ids
    assert np.abs(pr2["J0"].T @ pr2["J0"] - Ap).max() < 1e-8 * np.abs(Ap).max()
    # a constant plane stays out of the solve but stays in the prior (the marginalisation evaluates every block of its factors)
    scn_c, snap_c = plane_window(anchor=False, const=1)
    wc = oracle.solve(snap_c, abi.MARGIN_OLD)
    assert np.array_equal(wc["state"]["plane_R"], snap_c["plane_R"]) and wc["state"]["plane_Z"] == snap_c["plane_Z"]
    assert abi.BLK_PLANE_R in wc["prior"]["block_id"].tolist()
