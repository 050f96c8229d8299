"""GNSS inside the window solve and the marginalisation on the GPU (gfbe_window.gnss_ready; estimator.cpp:2965-3002, 3239-3291,
3459-3496, 3561-3590) against the CPU oracle, through the C ABI.

Tolerances. A pseudo-range is ~2.5e7 m in a double (4e-9 m resolution) and is weighted by up to 50, the device's sin / cos /
atan2 differ from the host's in the last bit: a GNSS residual agrees with the oracle's to ~1e-6 absolute (tests/test_gpu_gnss.py
states 1e-5), so a cost of ~3e3 made of ~100 such residuals agrees to ~1e-7 relative instead of the 1e-9 of a window without
GNSS: check_solve's bounds are taken times GNSS_LOOSE = 10 here (measured on MI355X, tools/diag_scripts/gnss_diag.py: cost history 1.3e-9
relative, final cost 5e-11, poses 1e-11 m, clock biases 5e-9 m, anchor 4e-9 m: inside the plain bounds already); the priors'
normal equations are compared on PRIOR_LOOSE = 300 (A' 1.3e-9 relative measured; b' — the marginal cost's gradient at a state the
solve left before it had settled, cancelling numbers of the information's size — 1.5e-4 of its largest entry, plus what the difference of the
linearisation points explains: check_prior). The receiver clock biases are metres (1e5 m in size), the anchor is ECEF metres."""
import numpy as np
import pytest

from _gfbe_import import gf
import gnss_window_cases as gw
from test_gpu_parity import check_solve
from test_gpu_plane import check_prior

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu
GNSS_LOOSE, PRIOR_LOOSE = 10.0, 300.0      # (PRIOR_LOOSE 100 until round 3; the 7 x 7 panel sums of round 4 end the early-stopping GNSS windows 1.5e-4 of |b'| apart)


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


def check_gnss_state(want, got, loose=1.0, yaw_free=False):
    a, b = want["state"]["gnss_state"], got["state"]["gnss_state"]
    if yaw_free:      # (a window that is not gnss_ready does not hold the yaw constant)
        assert abs(b["yaw_enu_local"] - a["yaw_enu_local"]) < loose * 1e-10
        b = dict(b, yaw_enu_local=a["yaw_enu_local"])
    assert np.abs(b["rcv_dt"] - a["rcv_dt"]).max() < loose * 1e-6          # metres
    assert np.abs(b["rcv_ddt"] - a["rcv_ddt"]).max() < loose * 1e-7         # metres / second
    assert np.abs(b["anc_ecef"] - a["anc_ecef"]).max() < loose * 1e-6       # metres, on 6.4e6
    assert b["yaw_enu_local"] == a["yaw_enu_local"]                         # constant in the solve, wrapped the same way


def test_gnss_factor_cost_and_first_linearisation(be, oracle):
    _, _, snap = gw.gnss_window(seed=83, L=60, n_per_frame=5)
    want, got = oracle.eval_factors(snap, robustify=True), be.eval_factors(snap, robustify=True)
    assert abs(got["cost"] - want["cost"]) < 1e-9 * want["cost"]
    base = dict(snap)
    base.pop("gnss")
    assert want["cost"] - oracle.eval_factors(base, robustify=True)["cost"] > 1.0          # (the GNSS factors are in the objective)


@pytest.mark.parametrize("seed,anchor,n_per_frame", [(81, False, 8), (85, True, 3)])
def test_gnss_window_solve_and_both_marginalisations(be, oracle, seed, anchor, n_per_frame):
    scn, tru, snap = gw.gnss_window(seed=seed, L=150, n_per_frame=n_per_frame, anchor=anchor)
    want, got = check_solve(be, oracle, snap, abi.MARGIN_OLD, loose=GNSS_LOOSE)
    check_gnss_state(want, got)
    assert want["summary"]["iterations"] >= 3 and want["summary"]["final_cost"] < 1e-3 * want["summary"]["initial_cost"]
    ids = got["prior"]["block_id"].tolist()
    for bid in [abi.BLK_RCV_DT0 + k for k in range(4)] + [abi.BLK_RCV_DDT0, abi.BLK_YAW_ENU, abi.BLK_ANC_ECEF]:
        assert bid in ids                                   # rcv_dt[1] -> rcv_dt[0], rcv_ddt[1] -> rcv_ddt[0], yaw and anchor kept
    check_prior(want["prior"], got["prior"], loose=PRIOR_LOOSE)
    # the next window carries that prior: both marginalisation flavours
    nxt = gw.next_gnss_window(scn, tru, want, seed=seed)
    for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
        want2, got2 = check_solve(be, oracle, nxt, flag, loose=GNSS_LOOSE)
        check_gnss_state(want2, got2)
        check_prior(want2["prior"], got2["prior"], loose=PRIOR_LOOSE)
    assert abi.BLK_POSE0 + 9 not in got2["prior"]["block_id"].tolist() and abi.BLK_RCV_DT0 in got2["prior"]["block_id"].tolist()


def test_slow_window_and_prior_without_gnss_ready(be, oracle):
    # lowspeed (estimator.cpp:2973-2981): no GNSS residual blocks in the solve, the frame-0 factors still marginalised
    _, _, snap = gw.gnss_window(seed=86, L=100, n_per_frame=4)
    slow = dict(snap)
    slow["speed_bias"] = np.array(snap["speed_bias"], float).copy()
    slow["speed_bias"][:, :2] *= 0.2
    want, got = check_solve(be, oracle, slow, abi.MARGIN_OLD)
    assert got["state"]["gnss_state"]["rcv_dt"].tolist() == np.asarray(snap["gnss_state"]["rcv_dt"]).tolist()
    check_prior(want["prior"], got["prior"], loose=PRIOR_LOOSE)
    assert abi.BLK_ANC_ECEF in got["prior"]["block_id"].tolist()
    # a prior with GNSS blocks in a window that is not gnss_ready: the blocks are free parameters of the prior factor alone
    scn, tru, snap = gw.gnss_window(seed=89, L=100, n_per_frame=4)
    res = oracle.solve(snap, abi.MARGIN_OLD)
    nxt = gw.next_gnss_window(scn, tru, res, seed=89)
    nxt.pop("gnss")
    want, got = check_solve(be, oracle, nxt, abi.MARGIN_SECOND_NEW, loose=GNSS_LOOSE)
    check_gnss_state(want, got, yaw_free=True)
    assert got["state"]["gnss_state"]["yaw_enu_local"] != nxt["gnss_state"]["yaw_enu_local"]


def test_gnss_windows_in_a_batch(be, oracle):
    """A GNSS window, a plain window and a second GNSS window in one batch: every window as the oracle solves it alone, and the
    batch bit-identical from run to run (no atomics in the GNSS sums)."""
    snaps = [gw.gnss_window(seed=91, L=120, n_per_frame=6)[2], synth.Scenario(seed=92, n_landmarks=150, use_wheel=True).window(0),
             gw.gnss_window(seed=93, L=90, n_per_frame=10, anchor=True)[2]]
    runs = []
    for _ in range(2):
        batch = be.batch_upload(snaps)
        batch.solve(abi.MARGIN_OLD)
        runs.append(batch.download())
        batch.free()
    for k, snap in enumerate(snaps):
        want = oracle.solve(snap, abi.MARGIN_OLD)
        got = runs[0][k]
        loose = GNSS_LOOSE if "gnss" in snap else 1.0
        assert got["summary"]["accepted"] == want["summary"]["accepted"]
        np.testing.assert_allclose(got["summary"]["cost_history"], want["summary"]["cost_history"], rtol=loose * 1e-6)
        assert np.abs(got["state"]["pose"] - want["state"]["pose"]).max() < loose * 1e-8
        check_prior(want["prior"], got["prior"], loose=PRIOR_LOOSE if "gnss" in snap else 1.0)
        assert got["summary"]["cost_history"] == runs[1][k]["summary"]["cost_history"]
        assert np.array_equal(got["state"]["pose"], runs[1][k]["state"]["pose"]) and np.array_equal(got["prior"]["J0"], runs[1][k]["prior"]["J0"])


def test_gnss_windows_in_a_throughput_batch(be, oracle):
    """34 windows (>= 32: the throughput kernel set — k_visasm, k_candidate_window, the dense factors on the side stream) of which
    every third carries GNSS: each window as the small-batch path solves it alone (tolerance: the two kernel sets sum the Schur
    partials in different groups), the GNSS state included, and the batch bit-identical from run to run."""
    kinds = [gw.gnss_window(seed=95, L=100, n_per_frame=5)[2], synth.Scenario(seed=96, n_landmarks=120, use_wheel=True).window(0),
             gw.gnss_window(seed=97, L=80, n_per_frame=9, anchor=True)[2]]
    snaps = [kinds[i % 3] for i in range(34)]
    runs = []
    for _ in range(2):
        batch = be.batch_upload(snaps)
        batch.solve(abi.MARGIN_OLD)
        runs.append(batch.download())
        batch.free()
    alone = [be.solve(k, abi.MARGIN_OLD) for k in kinds]
    for i, got in enumerate(runs[0]):
        want = alone[i % 3]
        assert got["summary"]["accepted"] == want["summary"]["accepted"] and got["summary"]["iterations"] == want["summary"]["iterations"]
        np.testing.assert_allclose(got["summary"]["cost_history"], want["summary"]["cost_history"], rtol=1e-7)
        assert np.abs(got["state"]["pose"] - want["state"]["pose"]).max() < 1e-8
        if "gnss" in snaps[i]:
            check_gnss_state(want, got)
        assert got["prior"]["block_id"].tolist() == want["prior"]["block_id"].tolist()
        assert got["summary"]["cost_history"] == runs[1][i]["summary"]["cost_history"] and np.array_equal(got["prior"]["J0"], runs[1][i]["prior"]["J0"])
        assert got["summary"]["cost_history"] == runs[0][i % 3]["summary"]["cost_history"]        # (independent of the slot inside the batch)


def test_gnss_window_beyond_the_lds_staging_limit(be, oracle):
    """Four constellations, 32 satellites per frame: 352 observations > GN_LDS_OBS = 320 (gfbe_gnss_solve.hip), so k_gnss takes the
    path that sums entry by entry from the global J / r copy — in the solve (mode 0) and in MARGIN_OLD (mode 2). ADVICE round 3."""
    scn, tru, snap = gw.gnss_window(seed=87, L=150, n_per_frame=32)
    assert len(snap["gnss"]["obs"]) > 320
    want, got = check_solve(be, oracle, snap, abi.MARGIN_OLD, loose=GNSS_LOOSE)
    check_gnss_state(want, got)
    check_prior(want["prior"], got["prior"], loose=PRIOR_LOOSE)
    # the same window inside a batch whose other window stages in LDS (the launch sizes the LDS for the batch's largest window, capped)
    _, _, small = gw.gnss_window(seed=81, L=150, n_per_frame=8)
    both = be.solve_batch([small, snap], abi.MARGIN_OLD)
    assert both[1]["summary"]["final_cost"] == got["summary"]["final_cost"]
    assert np.array_equal(both[1]["state"]["pose"], got["state"]["pose"])


def test_throughput_batch_with_priors_beyond_the_lds_staging_limit(be, oracle):
    """k_prior_tp stages J0 in LDS up to n = 90; a prior that carries GNSS blocks (clock biases, yaw, anchor) is larger, and a batch
    that holds one takes the global-memory path for ALL its priors. 33 windows (the throughput kernel set): second GNSS windows with
    the carried prior next to a plain window with its 86-dim prior, each as the small-batch path solves it alone."""
    scn, tru, snap = gw.gnss_window(seed=81, L=120, n_per_frame=6)
    first = oracle.solve(snap, abi.MARGIN_OLD)
    big = gw.next_gnss_window(scn, tru, first, seed=81)
    assert big["prior"]["n"] > 90
    ps = synth.Scenario(seed=98, n_landmarks=140, use_wheel=True)
    r0 = be.solve(ps.window(0), abi.MARGIN_OLD)
    plain = ps.window(1, state=synth.shift_state_for_next_window(ps, r0["state"], 1), prior=r0["prior"])
    assert 0 < plain["prior"]["n"] <= 90
    kinds = [big, plain]
    alone = [be.solve(k, abi.MARGIN_OLD) for k in kinds]
    batch = be.solve_batch([kinds[i % 2] for i in range(33)], abi.MARGIN_OLD)
    for i, got in enumerate(batch):
        want = alone[i % 2]
        assert got["summary"]["accepted"] == want["summary"]["accepted"] and got["summary"]["iterations"] == want["summary"]["iterations"]
        np.testing.assert_allclose(got["summary"]["cost_history"], want["summary"]["cost_history"], rtol=1e-7)
        assert np.abs(got["state"]["pose"] - want["state"]["pose"]).max() < 1e-8
        assert got["prior"]["block_id"].tolist() == want["prior"]["block_id"].tolist()
        assert got["summary"]["cost_history"] == batch[i % 2]["summary"]["cost_history"]
    # the same plain window in a batch of its own (its prior staged in LDS): the tolerances of the two kernel sets again
    staged = be.solve_batch([plain] * 32, abi.MARGIN_OLD)[5]
    np.testing.assert_allclose(staged["summary"]["cost_history"], alone[1]["summary"]["cost_history"], rtol=1e-7)
    assert np.abs(staged["state"]["pose"] - alone[1]["state"]["pose"]).max() < 1e-8


def _kernel_backend(kernel, iters=8):
    o = abi.default_options()
    o.solve_kernel, o.max_num_iterations = kernel, iters
    return gf.Backend(device=0, options=o)


def test_chain_kernel_with_gnss_columns_against_the_blocked_factorisation(oracle):
    """Round 6: a batch with GNSS blocks takes k_solve_chain_wide (the speed-bias chain eliminated first, the 58 GNSS dims as dense
    columns: gfbe_options.solve_kernel = 0) where rounds 3-5 took the blocked out-of-LDS factorisation k_solve_big (solve_kernel = 4).
    Same Gauss-Newton step entry by entry after ONE iteration, the same Cauchy direction bit for bit (same scaling, same gradient), and
    whole solves with the same accept / reject sequence — alone, and as windows of a throughput batch next to a window without GNSS."""
    cases = [gw.gnss_window(seed=81, L=150, n_per_frame=8)[2], gw.gnss_window(seed=85, L=150, n_per_frame=3, anchor=True)[2]]
    scn, tru, snap = gw.gnss_window(seed=81, L=120, n_per_frame=6)
    cases.append(gw.next_gnss_window(scn, tru, oracle.solve(snap, abi.MARGIN_OLD), seed=81))      # (carries the ~95-dim prior with GNSS blocks)
    for snap in cases:
        ys = []
        for kernel in (4, 0):
            be = _kernel_backend(kernel, iters=1)
            b = be.batch_upload([snap])
            b.solve(abi.MARGIN_NONE)
            ys.append((b.debug_vector(0), b.debug_vector(1), b.download()[0]["summary"]))
            b.free()
            be.close()
        (y1, v1, s1), (y0, v0, s0) = ys
        assert np.abs(y1).max() > 0.0 and np.abs(y1[abi.DENSE_DIM - 58:]).max() > 0.0      # (the GNSS dims take part in the step)
        assert np.array_equal(v0, v1)
        assert np.abs(y0 - y1).max() < 1e-7 * max(np.abs(y1).max(), 1.0), (np.abs(y0 - y1).max(), np.abs(y1).max())
        assert s0["accepted"] == s1["accepted"] and abs(s0["final_cost"] - s1["final_cost"]) < 1e-7 * s1["final_cost"]
    big, wide = _kernel_backend(4), _kernel_backend(0)
    plain = synth.Scenario(seed=96, n_landmarks=120, use_wheel=True).window(0)
    for snap in cases:
        want, a = check_solve(big, oracle, snap, abi.MARGIN_OLD, loose=GNSS_LOOSE)
        _, b = check_solve(wide, oracle, snap, abi.MARGIN_OLD, loose=GNSS_LOOSE)
        assert a["summary"]["accepted"] == b["summary"]["accepted"] and a["summary"]["termination"] == b["summary"]["termination"]
        assert abs(a["summary"]["final_cost"] - b["summary"]["final_cost"]) < 1e-8 * a["summary"]["final_cost"]
        check_gnss_state(a, b)
    batch = [cases[i % 3] if i % 4 else plain for i in range(36)]
    ra, rb = big.solve_batch(batch, abi.MARGIN_OLD), wide.solve_batch(batch, abi.MARGIN_OLD)
    for a, b in zip(ra, rb):
        assert a["summary"]["accepted"] == b["summary"]["accepted"] and a["summary"]["iterations"] == b["summary"]["iterations"]
        np.testing.assert_allclose(b["summary"]["cost_history"], a["summary"]["cost_history"], rtol=1e-7)
        assert np.abs(a["state"]["pose"] - b["state"]["pose"]).max() < 1e-8
    big.close(); wide.close()


def test_gnss_windows_at_and_beyond_nine_tile_columns_of_dense_dims(be, oracle):
    """Every optional block free (camera and wheel extrinsics, the wheel intrinsics, both time offsets) on top of the GNSS blocks: 141
    dense dims + the right-hand side — the last column but one of k_solve_chain_wide's nine tile columns. With the plane blocks on top
    (PlaneFactor on every pose: + 4 dims) the dense part no longer fits and the batch keeps the blocked factorisation — the structure
    check of the upload (gfbe_host.cpp: solve_chain_wide_fits). Both against the oracle like any window (the window with everything free
    stops before it has settled: the multiple of tests/test_gpu_branches.py::test_all_blocks_free_with_subset_masks)."""
    from test_gpu_branches import all_free
    _, _, snap = gw.gnss_window(seed=81, L=150, n_per_frame=8)
    free = all_free(snap, masks=False)
    want, got = check_solve(be, oracle, free, abi.MARGIN_OLD, loose=100.0 * GNSS_LOOSE)
    check_gnss_state(want, got, loose=100.0)
    rng = np.random.default_rng(5)
    q = synth.so3_exp(rng.normal(0, 0.01, 3))
    q[2] = 0.0
    plane = dict(free)
    plane["plane_R"] = q / np.linalg.norm(q)
    plane["plane_Z"] = -float(plane["ex_pose_wheel"][2]) + 0.02
    plane["plane"] = dict(noise_inv=[100.0, 100.0, 50.0], const=0)
    want, got = check_solve(be, oracle, plane, abi.MARGIN_OLD, loose=100.0 * GNSS_LOOSE)
    check_gnss_state(want, got, loose=100.0)
