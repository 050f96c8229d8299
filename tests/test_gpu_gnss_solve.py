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
