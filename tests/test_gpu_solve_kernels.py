"""The three factorisations of the reduced system (gfbe_options.solve_kernel; the DENSE_SCHUR linear solve of
estimator.cpp:3364-3379 after the landmark elimination): k_solve_chain (2) eliminates the speed-bias blocks as a chain of 9 x 9
blocks from one end before the dense pose / extrinsic part, k_solve_chain_tw (3; the default of batches below 32 windows) from
both ends at once — two chain waves that meet in the middle block —, k_solve (1) factorises everything as one tiled matrix. Same
Gauss-Newton step up to rounding — compared entry by entry through gfbe_debug_vector — on the window shapes that change the code
path: five and six tile columns of the dense part, a window that is still filling up, no speed-bias blocks at all; and the structure
check that hands a prior with a second speed-bias block to the monolithic kernel."""
import numpy as np
import pytest

from _gfbe_import import gf
from test_gpu_branches import all_free
from test_gpu_parity import check_solve, window_with_prior

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


def _backend(kernel, iters=8):
    o = abi.default_options()
    o.solve_kernel, o.max_num_iterations = kernel, iters
    return gf.Backend(device=0, options=o)


def _cases(oracle):
    _, with_prior = window_with_prior(oracle, 81, 600)
    scn = synth.Scenario(seed=82, n_landmarks=300, use_wheel=True)
    first = scn.window(0)
    _, free = window_with_prior(oracle, 83, 400)
    free = all_free(free)                                    # camera extrinsic + td free: 83 dense dims = six tile columns
    partial = dict(synth.Scenario(seed=84, n_landmarks=300, use_wheel=True).window(0))     # a window that is still filling up
    fc = 6
    keep = partial["vis_imu_j"] <= fc
    for k in list(partial):
        if k.startswith("vis_"):
            partial[k] = partial[k][keep]
    partial["frame_count"] = fc
    for k in ("imu", "imu_frame", "wheel", "wheel_frame"):
        partial[k] = partial[k][:fc]
    no_imu = dict(synth.Scenario(seed=85, n_landmarks=300, use_wheel=True).window(0))
    no_imu["imu"], no_imu["imu_frame"] = np.zeros((0, abi.IMU_DOUBLES)), np.zeros(0, np.int32)
    pc = np.zeros(abi.NFRAMES, np.uint8)
    pc[0] = 1
    no_imu["pose_const"] = pc
    return [("prior", with_prior), ("first", first), ("all_free", free), ("no_imu", no_imu), ("partial", partial)]


def test_first_gauss_newton_step_agrees_entry_by_entry(oracle):
    """One iteration from identical inputs: y (Jacobi-scaled Gauss-Newton step of the dense block), the Cauchy direction and the
    cost after the step."""
    for name, snap in _cases(oracle):
        ys = []
        for kernel in (1, 2, 3):
            be = _backend(kernel, iters=1)
            b = be.batch_upload([snap])
            b.solve(abi.MARGIN_NONE)
            ys.append((b.debug_vector(0), b.debug_vector(1), b.download()[0]["summary"]))
            b.free()
            be.close()
        y1, v1, s1 = ys[0]
        for kernel, (y0, v0, s0) in zip((2, 3), ys[1:]):
            assert np.array_equal(v0, v1), (name, kernel)                             # (same scaling, same gradient)
            assert np.abs(y0 - y1).max() < 1e-7 * max(np.abs(y1).max(), 1.0), (name, kernel, np.abs(y0 - y1).max(), np.abs(y1).max())
            assert s0["accepted"] == s1["accepted"], (name, kernel)
            assert abs(s0["final_cost"] - s1["final_cost"]) < 1e-7 * s1["final_cost"], (name, kernel)


@pytest.mark.parametrize("name", ["prior", "first", "all_free", "no_imu", "partial"])
def test_whole_solves_agree_and_match_the_oracle(oracle, name):
    snap = dict(_cases(oracle))[name]
    # (the window with every block free and subset masks stops on rejected steps before it has settled: the multiple of
    #  tests/test_gpu_branches.py::test_all_blocks_free_with_subset_masks)
    loose = 100.0 if name == "all_free" else 1.0
    res = []
    for kernel in (1, 2, 3):
        be = _backend(kernel)
        want, got = check_solve(be, oracle, snap, abi.MARGIN_OLD, loose=loose)
        res.append(got)
        be.close()
    a = res[0]
    for b in res[1:]:
        assert a["summary"]["accepted"] == b["summary"]["accepted"] and a["summary"]["termination"] == b["summary"]["termination"]
        assert abs(a["summary"]["final_cost"] - b["summary"]["final_cost"]) < loose * 2e-9 * a["summary"]["final_cost"]


def test_the_default_takes_the_two_ended_chain_below_32_windows_and_the_one_ended_one_from_there(oracle):
    """gfbe_options.solve_kernel = 0: a batch below 32 windows is solved bit for bit like solve_kernel = 3, a larger one like 2."""
    _, snap = window_with_prior(oracle, 87, 300)
    got = {}
    for kernel in (0, 2, 3):
        be = _backend(kernel)
        got[kernel] = (be.solve(snap, abi.MARGIN_OLD), be.solve_batch([snap] * 33, abi.MARGIN_OLD)[32])
        be.close()
    def same(a, b):
        return np.array_equal(a["state"]["pose"], b["state"]["pose"]) and np.array_equal(a["feature"], b["feature"]) and a["summary"] == b["summary"]
    assert same(got[0][0], got[3][0])           # one window: the two-ended chain
    assert same(got[0][1], got[2][1])           # 33 windows: the one-ended chain


def test_prior_with_a_second_speed_bias_block_takes_the_monolithic_kernel(oracle):
    """The reference's priors keep SpeedBias[0] only (estimator.cpp:3400-3433); the ABI allows any block table. A prior whose
    speed-bias block is relabelled SpeedBias[2] couples blocks the chain does not link: the batch must fall back to k_solve
    and still match the oracle, alone and next to ordinary windows."""
    _, snap = window_with_prior(oracle, 86, 300)
    odd = dict(snap)
    pr = dict(snap["prior"])
    ids = pr["block_id"].copy()
    assert abi.BLK_SB0 in ids.tolist()
    ids[ids.tolist().index(abi.BLK_SB0)] = abi.BLK_SB0 + 2
    pr["block_id"] = ids
    odd["prior"] = pr
    be = _backend(0)
    want, got = check_solve(be, oracle, odd, abi.MARGIN_NONE)
    both = be.solve_batch([snap, odd], abi.MARGIN_NONE)
    alone = be.solve(snap, abi.MARGIN_NONE)
    assert both[1]["summary"] == got["summary"]
    # (the ordinary window of the mixed batch went through k_solve too: same step to rounding, not to the bit)
    assert both[0]["summary"]["accepted"] == alone["summary"]["accepted"]
    assert abs(both[0]["summary"]["final_cost"] - alone["summary"]["final_cost"]) < 2e-9 * alone["summary"]["final_cost"]
    be.close()
