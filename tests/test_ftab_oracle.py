"""CPU pins of the feature-table oracle (oracle/gfo_ftab.cpp) — SURVEY.md §8c (iv): exact integer bookkeeping for
scripted and random add / remove / slide sequences, against hand-computed expectations and against an independent
pure-Python model (tests/ftab_model.py); triangulation against exact geometry."""
import numpy as np
import pytest

import ftab_model as fm
from _gfbe_import import gf

abi = gf.abi


@pytest.fixture()
def table(oracle):
    t = abi.FeatureTables(oracle.lib, "gfo_", None, n_tables=1, capacity=4096)
    yield fm.OneTable(t)
    t.close()


def rows(n, seed):
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.normal(0, 0.2, (n, 2)), np.ones(n), np.zeros((n, 5))])


def test_scripted_add_slide_sequence_exact_triples(table, oracle):
    """Hand-scripted tracker output; expectations written out by hand from feature_manager.cpp:57-88, 858-934 and
    the factor loop estimator.cpp:3330-3358."""
    # frames 0..10: ids 1,2 seen from frame 0; id 3 from frame 2; id 4 only in frames 3..5 (3 obs: never a landmark);
    # id 5 from frame 7
    for fc in range(11):
        ids = [1, 2] + ([3] if fc >= 2 else []) + ([4] if 3 <= fc <= 5 else []) + ([5] if fc >= 7 else [])
        kf, cnt, _ = table.add_frame(fc, ids, rows(len(ids), fc), 0.0)
        assert kf is True                      # fewer than 20 tracked features: always a keyframe
        if fc == 0:
            assert cnt == (0, 2, 0)
        if fc == 3:
            assert cnt == (3, 1, 2)            # ids 1,2,3 tracked, id 4 new, ids 1,2 reach 4 observations
    s = table.snapshot()
    assert s["feature_id"].tolist() == [1, 2, 3, 4, 5]          # insertion order
    assert s["start_frame"].tolist() == [0, 0, 2, 3, 7]
    assert s["n_obs"].tolist() == [11, 11, 9, 3, 4]
    fl = abi.ftab_to_feature_list(s)
    fl["estimated_depth"] = np.full(5, 2.0)
    f = oracle.build_visual_factors(fl)
    # landmarks = features with >= 4 observations, in list order: ids 1,2,3,5 -> feature_index 0..3
    want = [(0, 0, j) for j in range(1, 11)] + [(1, 0, j) for j in range(1, 11)] + [(2, 2, j) for j in range(3, 11)] + \
           [(3, 7, j) for j in range(8, 11)]
    assert list(zip(f["vis_feature_index"].tolist(), f["vis_imu_i"].tolist(), f["vis_imu_j"].tolist())) == want
    # MARGIN_OLD without depth hand-over: removeBack
    table.remove_back()
    s = table.snapshot()
    assert s["feature_id"].tolist() == [1, 2, 3, 4, 5]
    assert s["start_frame"].tolist() == [0, 0, 1, 2, 6]
    assert s["n_obs"].tolist() == [10, 10, 9, 3, 4]
    # the next frame arrives at frame_count = 10 and is NOT a keyframe in this script: removeFront(10)
    table.add_frame(10, [1, 3, 5], rows(3, 99), 0.0)
    s = table.snapshot()
    assert s["n_obs"].tolist() == [11, 10, 10, 3, 5]
    table.remove_front(10)
    s = table.snapshot()
    # id 1: start 0, end 10 >= 9 -> observation j = 9 erased; id 2: end 9 >= 9 -> obs 9 erased; id 3: start 1 -> obs 8 erased;
    # id 4 (frames 2..4): end 4 < 9 untouched; id 5: start 6 -> obs 3 erased
    assert s["n_obs"].tolist() == [10, 9, 9, 3, 4]
    assert s["start_frame"].tolist() == [0, 0, 1, 2, 6]
    f = oracle.build_visual_factors(abi.ftab_to_feature_list(dict(s, estimated_depth=np.full(5, 2.0))))
    want = [(0, 0, j) for j in range(1, 10)] + [(1, 0, j) for j in range(1, 9)] + [(2, 1, j) for j in range(2, 10)] + \
           [(3, 6, j) for j in range(7, 10)]
    assert list(zip(f["vis_feature_index"].tolist(), f["vis_imu_i"].tolist(), f["vis_imu_j"].tolist())) == want


def test_remove_back_shift_depth_rules(table):
    """feature_manager.cpp:818-856: start != 0 -> start-1; start == 0: first observation dropped, the feature dies with
    fewer than 2 left, otherwise the depth moves to the new first frame (INIT_DEPTH when it falls behind the camera)."""
    ob = lambda x, y: [x, y, 1.0, 0, 0, 0, 0, 0]
    table.add_frame(0, [10, 11, 12], [ob(0.1, 0.0), ob(0.0, 0.1), ob(0.2, 0.2)], 0.0)
    table.add_frame(1, [10, 11, 13], [ob(0.1, 0.0), ob(0.0, 0.1), ob(0.3, 0.0)], 0.0)
    table.add_frame(2, [10, 11, 13], [ob(0.1, 0.0), ob(0.0, 0.1), ob(0.3, 0.0)], 0.0)
    table.set_depth([])                       # nobody has 4 observations yet: no change
    s = table.snapshot()
    assert s["estimated_depth"].tolist() == [-1.0] * 4
    # give depths by hand through a 4-observation detour is not possible here; use clear + direct check of the formulas
    # with depth -1: pts_i = -uv, camera moved 1 m forward (new_P = (0,0,1)), identity rotations
    table.remove_back_shift_depth(np.eye(3), np.zeros(3), np.eye(3), np.array([0.0, 0.0, 1.0]))
    s = table.snapshot()
    assert s["feature_id"].tolist() == [10, 11, 13]          # id 12 had 1 observation -> erased; 10, 11 keep 2
    assert s["start_frame"].tolist() == [0, 0, 0]
    assert s["n_obs"].tolist() == [2, 2, 2]
    # dep_j = (-1)*1 - 1 = -2 <= 0 -> INIT_DEPTH for ids 10, 11; id 13 (start 1) untouched
    assert s["estimated_depth"].tolist() == [5.0, 5.0, -1.0]
    table.remove_back_shift_depth(np.eye(3), np.zeros(3), np.eye(3), np.array([0.0, 0.0, 1.0]))
    s = table.snapshot()
    assert s["feature_id"].tolist() == []                    # everybody is down to 1 observation -> all erased


def test_set_depth_failures_and_depth_vector(table):
    for fc in range(5):
        table.add_frame(fc, [1, 2, 3], rows(3, fc), 0.0)
    table.add_frame(5, [2, 3, 4], rows(3, 5), 0.0)
    table.set_depth([0.5, -0.25, 0.1])               # ids 1,2,3 have >= 4 observations; id 4 not
    s = table.snapshot()
    assert s["estimated_depth"].tolist() == [2.0, -4.0, 10.0, -1.0]
    assert s["solve_flag"].tolist() == [1, 2, 1, 0]
    np.testing.assert_array_equal(table.depth_vector(), [0.5, -0.25, 0.1])
    table.remove_failures()
    assert table.snapshot()["feature_id"].tolist() == [1, 3, 4]
    table.clear_depth()
    assert table.snapshot()["estimated_depth"].tolist() == [-1.0, -1.0, -1.0]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sequences_match_python_model(oracle, seed):
    t = abi.FeatureTables(oracle.lib, "gfo_", None, n_tables=1, capacity=4096)
    a, b = fm.PyFeatureManager(), fm.OneTable(t)
    log = fm.drive(np.random.default_rng(seed), [a, b], n_steps=45)
    for la, lb in log:
        assert la[0] == lb[0] and tuple(la[1]) == tuple(lb[1])
        assert abs(la[2] - lb[2]) <= 1e-12 * max(1.0, abs(la[2]))
    fm.assert_same_tables(a.snapshot(), b.snapshot(), depth_rtol=1e-14)
    assert len(a.feature) > 50
    t.close()


def test_triangulate_recovers_exact_depths(oracle):
    """Noise-free observations of known points from known poses: the SVD triangulation (feature_manager.cpp:669-724)
    returns the true depth in the start frame, 1e-9 relative; RGB-D triangulation (:726-799) averages verified depths."""
    rng = np.random.default_rng(5)
    scn = gf.synth.Scenario(seed=3, n_landmarks=10, use_wheel=False, noise=False)
    st = scn.truth_state(0)
    poses = abi.pose_rows(st["pose"])
    tic_ric = np.concatenate([scn.tic, scn.ric.ravel()])
    t = abi.FeatureTables(oracle.lib, "gfo_", None, n_tables=1, capacity=256, options=dict(depth_threshold=20.0))
    truth = {}
    frames = {fc: ([], []) for fc in range(11)}
    for fid in range(40):
        start, n = int(rng.integers(0, 6)), int(rng.integers(2, 6))
        depth = rng.uniform(1.5, 9.0)
        pn = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1.0])
        P, R = st["pose"][start, :3], poses[start, 3:].reshape(3, 3)
        pw = R @ (scn.ric @ (pn * depth) + scn.tic) + P
        truth[fid] = (depth, n)
        for j in range(start, start + n):
            Pj, Rj = st["pose"][j, :3], poses[j, 3:].reshape(3, 3)
            pc = scn.ric.T @ (Rj.T @ (pw - Pj) - scn.tic)
            frames[j][0].append(fid)
            frames[j][1].append([pc[0] / pc[2], pc[1] / pc[2], 1.0, 0, 0, 0, 0, pc[2]])
    for fc in range(11):
        t.add_frame([fc], [frames[fc][0]], [np.array(frames[fc][1]).reshape(-1, 8)], [0.0])
    t.triangulate([poses], [tic_ric])
    s = t.download(0)
    for k, fid in enumerate(s["feature_id"]):
        depth, n = truth[int(fid)]
        if n >= 4:
            assert abs(s["estimated_depth"][k] - depth) < 1e-9 * depth
            assert s["estimate_flag"][k] == 2
        else:
            assert s["estimated_depth"][k] == -1.0
    t.clear_depth()
    t.triangulate([poses], [tic_ric], with_depth=True)
    s = t.download(0)
    for k, fid in enumerate(s["feature_id"]):
        depth, n = truth[int(fid)]
        if n >= 4:
            assert abs(s["estimated_depth"][k] - depth) < 1e-9 * depth
            assert s["estimate_flag"][k] == 1
    # all reprojection errors are zero: nothing to reject in either mode
    assert len(t.check_outliers([poses], [tic_ric], 0)[0]) == 0
    assert len(t.check_outliers([poses], [tic_ric], 1)[0]) == 0
    t.close()


def test_slide_window_state(oracle):
    st = abi.State()
    for i in range(11):
        for k in range(7):
            st.para_Pose[i][k] = 100 * i + k
        for k in range(9):
            st.para_SpeedBias[i][k] = 1000 * i + k
    import ctypes as C
    a, b = abi.State(), abi.State()
    C.memmove(C.byref(a), C.byref(st), C.sizeof(st))
    C.memmove(C.byref(b), C.byref(st), C.sizeof(st))
    oracle.lib.gfo_slide_window_state(C.byref(a), abi.MARGIN_OLD)
    assert [a.para_Pose[i][0] for i in range(11)] == [100.0 * (i + 1) for i in range(10)] + [1000.0]
    assert [a.para_SpeedBias[i][8] for i in range(11)] == [1000.0 * (i + 1) + 8 for i in range(10)] + [10008.0]
    oracle.lib.gfo_slide_window_state(C.byref(b), abi.MARGIN_SECOND_NEW)
    assert [b.para_Pose[i][0] for i in range(11)] == [100.0 * i for i in range(9)] + [1000.0, 1000.0]
