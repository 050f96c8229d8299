"""gfbe_options.merge_lin_schur (round 6): throughput batches evaluate the visual factors AND eliminate the landmarks in one launch
(k_linschur: one workgroup per window and group of start frames, the landmark rows go from the evaluating lanes' registers into the LDS
panel of the Schur product) instead of k_vis<0, false> + k_schur exchanging every factor's row through HBM. Same quantities, the
per-landmark sums added in another order: compared with the two-kernel sequence at the tolerances the other pairs of kernel sets are
compared at (tests/test_gpu_parity.py::test_large_batch_throughput_path), with the oracle through check_solve's bounds, bit for bit with
itself (position in the batch, repeated solves, the parts of a split batch), and to rounding between the two launch sequences it is part
of (speculative_linearization on / off: the candidate's pass runs k_linschur<SPEC>, the other sequence k_linschur<false> every iteration),
including the in-kernel mu retry (whose slow E rebuild reads the rows k_linschur still writes for k_lm_step)."""
import numpy as np
import pytest

from _gfbe_import import gf
from test_gpu_parity import window_with_prior

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


def _backend(**kw):
    o = abi.default_options()
    o.merge_lin_schur = 1      # (the option is off by default: measured slower than the two kernels, include/gfbe.h)
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


def _close(a, b):
    sa, sb = a["summary"], b["summary"]
    assert sa["accepted"] == sb["accepted"] and sa["termination"] == sb["termination"] and sa["iterations"] == sb["iterations"]
    np.testing.assert_allclose(sb["cost_history"], sa["cost_history"], rtol=1e-7)
    assert abs(sa["final_cost"] - sb["final_cost"]) < 2e-8 * sa["final_cost"], (sa["final_cost"], sb["final_cost"])      # (the 40-landmark window without wheel or prior: 5e-9)
    assert np.abs(a["state"]["pose"] - b["state"]["pose"]).max() < 1e-8
    assert np.abs(a["state"]["speed_bias"] - b["state"]["speed_bias"]).max() < 1e-7
    np.testing.assert_allclose(b["feature"], a["feature"], rtol=1e-7, atol=1e-12)
    if a.get("prior") is not None:
        Aa, Ab = a["prior"]["J0"].T @ a["prior"]["J0"], b["prior"]["J0"].T @ b["prior"]["J0"]
        assert np.abs(Aa - Ab).max() < 1e-7 * np.abs(Aa).max()


def _cases(oracle):
    """Windows of several shapes: with / without prior and wheel, few and many landmarks (one to several tiles per start frame, start
    frames without a landmark), constant landmarks, a window that is still filling up."""
    _, w1 = window_with_prior(oracle, 191, 500)
    _, w2 = window_with_prior(oracle, 192, 150)
    plain = synth.Scenario(seed=193, n_landmarks=900, use_wheel=True).window(0)
    small = synth.Scenario(seed=194, n_landmarks=40, use_wheel=False).window(0)
    return [w1, w2, plain, small]


def test_merged_launch_against_the_two_kernels(oracle):
    snaps = _cases(oracle)
    big = [snaps[i % len(snaps)] for i in range(36)]
    got = {}
    for merge in (0, 1):
        be = _backend(merge_lin_schur=merge)
        got[merge] = be.solve_batch(big, abi.MARGIN_OLD)
        again = be.solve_batch(big, abi.MARGIN_OLD)
        be.close()
        for i, (a, b) in enumerate(zip(got[merge], again)):      # repeatable, and independent of the place in the batch
            assert _identical(a, b) and _identical(a, got[merge][i % len(snaps)])
    for a, b in zip(got[0][:len(snaps)], got[1][:len(snaps)]):
        _close(a, b)


def test_merged_launch_against_the_oracle(be, oracle):
    from test_gpu_parity import check_solve
    _, w1 = window_with_prior(oracle, 195, 700)
    want, single = check_solve(be, oracle, w1, abi.MARGIN_OLD)
    bm = _backend()
    many = bm.solve_batch([w1] * 33, abi.MARGIN_OLD)
    bm.close()
    for g in many:
        assert _identical(g, many[0])
    _close(single, many[0])
    assert many[0]["summary"]["accepted"] == want["summary"]["accepted"]
    assert abs(many[0]["summary"]["final_cost"] - want["summary"]["final_cost"]) < 1e-7 * want["summary"]["final_cost"]
    assert np.abs(many[0]["state"]["pose"][:, :3] - want["state"]["pose"][:, :3]).max() < 1e-7


@pytest.mark.parametrize("kw", [{}, {"test_fail_chol_iter": 2}, {"test_fail_chol_iter": 1, "test_fail_chol_count": 3}, {"max_num_iterations": 3},
                                {"max_num_iterations": 1}])
def test_both_launch_sequences(oracle, kw):
    """speculative_linearization off: k_linschur<false> in front of every iteration, the candidate's cost from the cost pass k_vis<1>; on:
    k_linschur<false> once, then k_linschur<SPEC> at every candidate (the gated launch in front of the later iterations). The two
    evaluate the same factors at the same states; what differs is the ORDER in which a tile's candidate cost is summed (k_vis<1>: per
    lane over its steps; k_linschur: per role, then over the roles), so — unlike the two-kernel sequences, which are bit-identical — the
    costs agree to rounding and everything discrete exactly."""
    snaps = _cases(oracle)
    big = [snaps[i % len(snaps)] for i in range(34)]
    got = []
    for spec in (0, 1):
        be = _backend(speculative_linearization=spec, **kw)
        got.append(be.solve_batch(big, abi.MARGIN_OLD) + be.solve_batch(big[:33], abi.MARGIN_SECOND_NEW))
        again = be.solve_batch(big, abi.MARGIN_OLD)
        assert all(_identical(a, b) for a, b in zip(got[-1], again))      # (each sequence repeats itself bit for bit)
        be.close()
    for a, b in zip(*got):
        _close(a, b)


def test_split_batch_and_constant_landmarks(oracle):
    """A batch solved as four parts side by side (each part its own launches), with constant landmarks (no row in the Schur panel) in
    some windows: bit for bit the batch solved whole."""
    snaps = _cases(oracle)
    fixed = dict(snaps[0])
    fc = np.zeros(len(fixed["para_feature"]), np.uint8)
    fc[::3] = 1
    fixed["feature_const"] = fc
    big = ([fixed] + snaps) * 28       # 140 windows: split_batch = 4 gives parts of 35
    got = []
    for split in (0, 4):
        be = _backend(split_batch=split)
        got.append(be.solve_batch(big, abi.MARGIN_OLD))
        be.close()
    assert all(_identical(a, b) for a, b in zip(*got))
    be = _backend(merge_lin_schur=0)
    ref = be.solve_batch(big[:35], abi.MARGIN_OLD)
    be.close()
    for a, b in zip(ref[:5], got[0][:5]):
        _close(a, b)
