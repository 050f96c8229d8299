"""gfbe_options.speculative_linearization (round 5): the pass that evaluates the candidate of a trust-region iteration linearises there, into
a second set of the linearisation's outputs, and the next iteration starts at the landmark elimination — TrustRegionMinimizer's own
order of evaluations (candidate cost, then residuals + Jacobians at the accepted point: the same state), one launch less per
iteration. Nothing a solve returns may change by a bit: compared here with the option off on the window shapes that take the
different paths — accepted and rejected steps, the mu retry of a failed factorisation (the weights of the landmark elimination are
formed with the mu of the iteration, not the one the candidate pass saw), a capped iteration count (the last candidate pass only
needs the costs), both kernel sets (one window; 33 windows), a batch with a free camera extrinsic."""
import numpy as np
import pytest

from _gfbe_import import gf
from test_gpu_branches import all_free
from test_gpu_parity import window_with_prior
from plane_cases import plane_window, next_plane_window

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


def _backend(spec, **kw):
    o = abi.default_options()
    o.speculative_linearization = spec
    for k, v in kw.items():
        setattr(o, k, v)
    return gf.Backend(device=0, options=o)


def _same(a, b):
    if isinstance(a, dict):
        return set(a) == set(b) and all(_same(a[k], b[k]) for k in a)
    if a is None or b is None:
        return a is b
    return np.array_equal(np.asarray(a), np.asarray(b))


def _identical(a, b):
    return _same(a["state"], b["state"]) and np.array_equal(a["feature"], b["feature"]) and a["summary"] == b["summary"] and _same(a.get("prior"), b.get("prior"))


def _windows(oracle):
    _, w1 = window_with_prior(oracle, 91, 500)
    _, w2 = window_with_prior(oracle, 92, 200)
    first = synth.Scenario(seed=93, n_landmarks=300, use_wheel=True).window(0)
    return w1, w2, first


@pytest.mark.parametrize("kw", [{}, {"test_fail_chol_iter": 2}, {"test_fail_chol_iter": 1, "test_fail_chol_count": 3}, {"max_num_iterations": 3},
                                {"max_num_iterations": 1}, {"solve_kernel": 1}, {"solve_kernel": 2}])
def test_single_window_is_unchanged_bit_for_bit(oracle, kw):
    w1, w2, first = _windows(oracle)
    got = []
    for spec in (0, 1):
        be = _backend(spec, **kw)
        got.append([be.solve(w, abi.MARGIN_OLD) for w in (w1, first)] + [be.solve(w2, abi.MARGIN_SECOND_NEW), be.solve(w1, abi.MARGIN_NONE)])
        be.close()
    assert all(_identical(a, b) for a, b in zip(*got))


def test_rejected_steps_keep_the_old_linearisation(oracle):
    """The window with every block free rejects steps (tests/test_gpu_branches.py), the plane window's successor too: a rejected candidate's
    linearisation is dropped and the next iteration reuses the current one (DoglegStrategy's reuse)."""
    _, w1 = window_with_prior(oracle, 94, 400)
    scn, pw = plane_window(anchor=True)
    be0 = _backend(0)
    nxt = next_plane_window(scn, pw, be0.solve(pw, abi.MARGIN_OLD))
    be0.close()
    cases = [all_free(w1), pw, nxt]
    got = []
    for spec in (0, 1):
        be = _backend(spec)
        got.append([be.solve(w, abi.MARGIN_OLD) for w in cases])
        be.close()
    assert any(0 in r["summary"]["accepted"][1:] for r in got[1])        # (the case is what it claims to be)
    assert all(_identical(a, b) for a, b in zip(*got))


def test_throughput_kernel_set_is_unchanged_bit_for_bit(oracle):
    """33 windows and more run the throughput kernels (k_vis, k_dense_tp, k_prior_tp, k_schur, k_visasm, k_solve_chain, k_lm_step): a plain
    batch, and one with a free camera extrinsic (the 20-column panel; its windows reject steps)."""
    w1, w2, first = _windows(oracle)
    for snaps in ([w1, w2, first] * 11, [w1, all_free(w2), first, w2] * 9):
        got = []
        for spec in (0, 1):
            be = _backend(spec)
            got.append(be.solve_batch(snaps, abi.MARGIN_OLD))
            be.close()
        assert all(_identical(a, b) for a, b in zip(*got))


def test_lidar_windows_are_unchanged_bit_for_bit(oracle):
    """Joint LIO + VIO windows (BASELINE configs[4]): the LiDAR factors' candidate cost is a launch of its own — k_lio_window linearises at the
    candidate like the visual and the dense factors. One window (the accept is a launch of its own there) and a batch of 33."""
    scn, w1 = window_with_prior(oracle, 95, 400)
    w1 = dict(w1, lio=synth.lidar_block(scn, 1, n=1500, seed=3, outliers=0.05))
    scn2 = synth.Scenario(seed=96, n_landmarks=300, use_wheel=True)
    w2 = dict(scn2.window(0), lio=synth.lidar_block(scn2, 0, n=5000, seed=6, frame=7, huber_delta=0.0))
    plain = scn2.window(0)
    got = []
    for spec in (0, 1):
        be = _backend(spec)
        got.append([be.solve(w1, abi.MARGIN_OLD), be.solve(w2, abi.MARGIN_SECOND_NEW)] + be.solve_batch([w1, plain, w2] * 11, abi.MARGIN_OLD))
        be.close()
    assert all(_identical(a, b) for a, b in zip(*got))


def test_gnss_windows_are_unchanged_bit_for_bit(oracle):
    """GNSS windows (gfbe_window.gnss_ready): the speculative pass evaluates the pseudo-range / Doppler / clock factors at the candidate
    (k_gnss, sub 1: per-observation J, r and the cost), the next iteration adds their sums into H (sub 2) — the operations of the one
    launch of the other sequence, in its order. One window, its successor (prior with GNSS blocks), a small batch and 33 windows."""
    import gnss_window_cases as gw
    scn, tru, g1 = gw.gnss_window(seed=97, L=120, n_per_frame=6)
    be0 = _backend(0)
    g2 = gw.next_gnss_window(scn, tru, be0.solve(g1, abi.MARGIN_OLD), seed=97)
    be0.close()
    g3 = gw.gnss_window(seed=98, L=90, n_per_frame=10, anchor=True)[2]
    plain = synth.Scenario(seed=99, n_landmarks=150, use_wheel=True).window(0)
    got = []
    for spec in (0, 1):
        be = _backend(spec)
        got.append([be.solve(g1, abi.MARGIN_OLD), be.solve(g2, abi.MARGIN_OLD), be.solve(g3, abi.MARGIN_SECOND_NEW)] +
                   be.solve_batch([g1, plain, g3], abi.MARGIN_OLD) + be.solve_batch([g1, plain, g3] * 11, abi.MARGIN_OLD))
        be.close()
    assert all(_identical(a, b) for a, b in zip(*got))


def _empty_and_partial():
    scn = synth.Scenario(seed=70, n_landmarks=100, use_wheel=True)
    partial = scn.window(0)                       # frame_count < WINDOW_SIZE: no marginalisation (estimator.cpp:3391)
    keep = partial["vis_imu_j"] <= 6
    for k in list(partial):
        if k.startswith("vis_"):
            partial[k] = partial[k][keep]
    partial["frame_count"] = 6
    partial["imu"], partial["imu_frame"] = partial["imu"][:6], partial["imu_frame"][:6]
    partial["wheel"], partial["wheel_frame"] = partial["wheel"][:6], partial["wheel_frame"][:6]
    empty = synth.Scenario(seed=71, n_landmarks=0, use_wheel=True).window(0)      # a window without a landmark: no tile of its own
    return empty, partial


def test_empty_and_partial_windows_inside_a_batch(oracle):
    """A window without landmarks (none of the batch's tiles is its own) and a partial window beside ordinary ones, in both kernel sets:
    the same bits with the option on and off, and the same bits as each window solved alone."""
    w1, w2, first = _windows(oracle)
    empty, partial = _empty_and_partial()
    alone_be = _backend(1)
    alone = [alone_be.solve(s, abi.MARGIN_OLD) for s in (empty, w1, partial, first, w2)]
    alone_be.close()
    for snaps, reps in (([empty, w1, partial, first, w2], 1), ([empty, w1, partial, first, w2] * 7, 7)):
        got = []
        for spec in (0, 1):
            be = _backend(spec)
            got.append(be.solve_batch(snaps, abi.MARGIN_OLD))
            be.close()
        assert all(_identical(a, b) for a, b in zip(*got))
        if reps == 1:      # (the small-batch kernel set is the single window's own: identical to the windows solved alone)
            assert all(_identical(a, b) for a, b in zip(got[1], alone))
