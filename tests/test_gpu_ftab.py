"""Device-resident feature tables (csrc/gfbe_ftab.hip) vs the CPU oracle (oracle/gfo_ftab.cpp), through the C ABI.
Integer contents and moved observations are bit-exact; depths: the hand-over of removeBackShiftDepth and setDepth to
1e-13 relative (same formulas), SVD triangulation to 1e-8 relative (4x4 Gram + Jacobi eigen on the device, one-sided
Jacobi SVD in the oracle)."""
import numpy as np
import pytest

import ftab_model as fm
from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return gf.Backend(device=0)


@pytest.mark.parametrize("seed", [11, 12])
def test_random_sequences_match_oracle(be, oracle, seed):
    dev = abi.FeatureTables(be.lib, "gfbe_", be.ctx, n_tables=1, capacity=4096)
    ref = abi.FeatureTables(oracle.lib, "gfo_", None, n_tables=1, capacity=4096)
    a, b = fm.OneTable(ref), fm.OneTable(dev)
    log = fm.drive(np.random.default_rng(seed), [a, b], n_steps=45)
    for la, lb in log:
        assert la[0] == lb[0] and tuple(la[1]) == tuple(lb[1])            # keyframe decision, counters
        assert abs(la[2] - lb[2]) <= 1e-12 * max(1.0, abs(la[2]))         # average parallax (summation order)
    fm.assert_same_tables(a.snapshot(), b.snapshot(), depth_rtol=1e-13)
    np.testing.assert_allclose(a.depth_vector(), b.depth_vector(), rtol=1e-13)
    assert len(a.snapshot()["feature_id"]) > 50
    dev.close(); ref.close()


def test_three_tables_in_one_launch(be, oracle):
    """W = 3 tables with different contents in every launch == three single-table oracles."""
    W = 3
    dev = abi.FeatureTables(be.lib, "gfbe_", be.ctx, n_tables=W, capacity=1024)
    refs = [abi.FeatureTables(oracle.lib, "gfo_", None, n_tables=1, capacity=1024) for _ in range(W)]
    rngs = [np.random.default_rng(100 + w) for w in range(W)]
    state = [dict(next_id=0, alive=[]) for _ in range(W)]
    for fc_step in range(16):
        fc = min(fc_step, 10)
        ids, obs = [], []
        for w in range(W):
            i, o, state[w]["alive"], state[w]["next_id"] = fm.random_frame(rngs[w], state[w]["next_id"], state[w]["alive"], 10 + 5 * w)
            ids.append(i); obs.append(o)
        kf, cnt, avg = dev.add_frame([fc] * W, ids, obs, [0.001 * w for w in range(W)])
        for w in range(W):
            k1, c1, a1 = refs[w].add_frame([fc], [ids[w]], [obs[w]], [0.001 * w])
            assert kf[w] == k1[0] and cnt[w].tolist() == c1[0].tolist() and abs(avg[w] - a1[0]) <= 1e-12 * max(1, abs(a1[0]))
        if fc < 10:
            continue
        xs = []
        for w in range(W):
            L = int((refs[w].download(0)["n_obs"] >= 4).sum())
            x = 1.0 / rngs[w].uniform(0.5, 9.0, L)
            x[rngs[w].random(L) < 0.1] *= -1
            xs.append(x)
            refs[w].set_depth([x]); refs[w].remove_failures()
        dev.set_depth(xs); dev.remove_failures()
        if fc_step % 2:
            PR = [np.concatenate([rngs[w].normal(0, 0.1, 3), np.eye(3).ravel()]) for w in range(W)]
            PN = [np.concatenate([rngs[w].normal(0, 0.1, 3), np.eye(3).ravel()]) for w in range(W)]
            dev.remove_back_shift_depth(PR, PN)
            for w in range(W):
                refs[w].remove_back_shift_depth([PR[w]], [PN[w]])
        else:
            dev.remove_front([10] * W)
            for w in range(W):
                refs[w].remove_front([10])
    assert dev.size().tolist() == [int(r.size()[0]) for r in refs]
    for w in range(W):
        fm.assert_same_tables(refs[w].download(0), dev.download(w), depth_rtol=1e-13)
    dev.close()
    for r in refs:
        r.close()


def test_triangulation_and_outlier_checks_match_oracle(be, oracle):
    scn = synth.Scenario(seed=31, n_landmarks=400, use_wheel=False)
    fl = scn.feature_list(0, extra_short=30)
    st = scn.initial_state(0)
    poses = abi.pose_rows(st["pose"])
    tic_ric = np.concatenate([scn.tic, scn.ric.ravel()])
    # replay the list as a tracker stream: feature k (id = k) observed in frames start .. start + n_obs - 1
    off = np.concatenate([[0], np.cumsum(fl["n_obs"])])
    frames = {fc: ([], []) for fc in range(11)}
    rng = np.random.default_rng(4)
    for k in range(len(fl["n_obs"])):
        for o in range(fl["n_obs"][k]):
            fc = fl["start_frame"][k] + o
            row = fl["obs"][off[k] + o]
            frames[fc][0].append(k)
            frames[fc][1].append(list(row) + [rng.uniform(0.05, 12.0)])
    tabs = [abi.FeatureTables(be.lib, "gfbe_", be.ctx, 1, 1024, options=dict(depth_threshold=10.0)),
            abi.FeatureTables(oracle.lib, "gfo_", None, 1, 1024, options=dict(depth_threshold=10.0))]
    for t in tabs:
        for fc in range(11):
            t.add_frame([fc], [frames[fc][0]], [np.array(frames[fc][1]).reshape(-1, 8)], [0.0])
        t.triangulate([poses], [tic_ric])
    a, b = tabs[1].download(0), tabs[0].download(0)
    fm.assert_same_tables(a, b, depth_rtol=1e-8)
    assert (a["estimate_flag"] == 2).sum() > 300
    # reprojection-error based rejection on the triangulated depths (some are poor: noisy initial poses)
    for mode in (0, 1):
        ra, rb = tabs[1].check_outliers([poses], [tic_ric], mode)[0], tabs[0].check_outliers([poses], [tic_ric], mode)[0]
        assert ra.tolist() == rb.tolist()
    assert len(tabs[1].check_outliers([poses], [tic_ric], 1)[0]) > 0
    # RGB-D flavour
    for t in tabs:
        t.clear_depth()
        t.triangulate([poses], [tic_ric], with_depth=True)
    fm.assert_same_tables(tabs[1].download(0), tabs[0].download(0), depth_rtol=1e-12)
    for t in tabs:
        t.close()


def test_capacity_overflow_fails_loudly(be):
    t = abi.FeatureTables(be.lib, "gfbe_", be.ctx, n_tables=1, capacity=8)
    with pytest.raises(RuntimeError):
        t.add_frame([0], [list(range(9))], [np.ones((9, 8))], [0.0])
    t.close()


def _fill_tables(tables, streams):
    for fc in range(abi.WINDOW_SIZE + 1):
        tables.add_frame([fc] * len(streams), [S.frames[fc][0] for S in streams], [S.frames[fc][1] for S in streams], [0.0] * len(streams))


def _stream_window(be, S):
    """The window snapshot of frames 0..10 of stream S without its visual part, the pose rows, [tic | ric]."""
    scn = S.scn
    Wn = abi.WINDOW_SIZE
    st = scn.initial_state(0)
    snap = dict(st)
    snap["frame_count"] = Wn
    snap["imu"] = be.preintegrate_imu([scn.imu_raw[i] for i in range(Wn)], scn.ba_est, scn.bg_est, (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W))
    snap["imu_frame"] = np.arange(Wn, dtype=np.int32)
    snap["wheel"] = be.preintegrate_wheel([scn.wheel_raw[i] for i in range(Wn)], [1.0, 1.0, 1.0, 0.0], (synth.VEL_N_WHEEL, synth.GYR_N_WHEEL))
    snap["wheel_frame"] = np.arange(Wn, dtype=np.int32)
    snap.update(ex_cam_const=1, ex_wheel_const=1, ix_wheel_const=1, td_const=1, td_wheel_const=1, prior=None)
    return snap, abi.pose_rows(st["pose"]), np.concatenate([scn.tic, scn.ric.ravel()])


def test_table_fed_batch_equals_host_upload(be):
    """gfbe_batch_upload_tables (landmark arrays packed on the device from the resident tables) == gfbe_batch_upload of
    the factors the host builds from the downloaded tables: same slots, same records, so every output is bit-identical."""
    W = 3
    stream = gf.stream
    tables = abi.FeatureTables(be.lib, "gfbe_", be.ctx, n_tables=W, capacity=4096, options=dict(min_parallax=14.0 / 600, depth_threshold=6.0))
    streams = [stream.Stream(seed=21 + w, n_kf=12, new_per_frame=40 + 25 * w, rgbd=(w == 1)) for w in range(W)]
    _fill_tables(tables, streams)
    snaps, poses, tr = [], [], []
    for w in range(W):
        s, p, t = _stream_window(be, streams[w])
        snaps.append(s); poses.append(p); tr.append(t)
    tables.triangulate(poses, tr, with_depth=True)         # table 1 holds measured depths: constant landmarks (estimate_flag 1)
    tables.triangulate(poses, tr, with_depth=False)
    full = []
    for w in range(W):
        tab = tables.download(w)
        assert (tab["estimate_flag"] == 1).any() == (w == 1)
        s = dict(snaps[w])
        s.update(be.build_visual_factors(abi.ftab_to_feature_list(tab)))
        full.append(s)
    assert len({len(s["para_feature"]) for s in full}) == W and min(len(s["para_feature"]) for s in full) > 100
    for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
        a = be.batch_upload(full)
        b = be.batch_upload_tables(tables, snaps)
        a.solve(flag); b.solve(flag)
        ra, rb = a.download(), b.download()
        a.free(); b.free()
        for x, y in zip(ra, rb):
            assert x["status"] == y["status"]
            assert x["summary"] == y["summary"]
            np.testing.assert_array_equal(x["feature"], y["feature"])
            fx, fy = abi.flat_state(x["state"]), abi.flat_state(y["state"])
            for k in fx:
                np.testing.assert_array_equal(np.asarray(fx[k]), np.asarray(fy[k]))
            assert (x["prior"] is None) == (y["prior"] is None) == (flag == abi.MARGIN_SECOND_NEW)   # no previous prior: nothing to carry
            for k in ("J0", "r0", "x0", "block_id", "block_idx") if x["prior"] else ():
                np.testing.assert_array_equal(x["prior"][k], y["prior"][k])
    # the solved depths go back into the tables exactly as through the host list
    tables.set_depth([r["feature"] for r in rb])
    for w in range(W):
        lm = tables.download(w)["n_obs"] >= 4
        np.testing.assert_array_equal(tables.download(w)["estimated_depth"][lm], 1.0 / rb[w]["feature"])
    tables.close()


def test_table_fed_batch_bad_inputs(be):
    tables = abi.FeatureTables(be.lib, "gfbe_", be.ctx, n_tables=1, capacity=1024)
    S = gf.stream.Stream(seed=5, n_kf=12, new_per_frame=5)
    _fill_tables(tables, [S])
    snap, _, _ = _stream_window(be, S)
    with pytest.raises(RuntimeError, match="more windows than tables"):
        be.batch_upload_tables(tables, [snap, snap])
    tables.close()


def test_table_fed_large_batch_two_halves(be):
    """130 tables -> the batch is split into two halves (tables 0..64 and 65..129) exactly as gfbe_batch_upload splits it."""
    W = 130
    base = [gf.stream.Stream(seed=31 + q, n_kf=12, new_per_frame=12 + 6 * q) for q in range(4)]
    streams = [base[w % 4] for w in range(W)]
    tables = abi.FeatureTables(be.lib, "gfbe_", be.ctx, n_tables=W, capacity=512)
    _fill_tables(tables, streams)
    win = [_stream_window(be, S) for S in base]
    snaps = [win[w % 4][0] for w in range(W)]
    tables.triangulate([win[w % 4][1] for w in range(W)], [win[w % 4][2] for w in range(W)], with_depth=False)
    full = []
    for w in range(W):
        s = dict(snaps[w])
        s.update(be.build_visual_factors(abi.ftab_to_feature_list(tables.download(w))))
        full.append(s)
    a, b = be.batch_upload(full), be.batch_upload_tables(tables, snaps)
    assert [be.lib.gfbe_batch_feature_count(b.h, w) for w in (0, 64, 65, 129, 130)] == [len(full[w]["para_feature"]) for w in (0, 64, 65, 129)] + [-1]
    a.solve(abi.MARGIN_OLD); b.solve(abi.MARGIN_OLD)
    ra, rb = a.download(), b.download()
    a.free(); b.free()
    for x, y in zip(ra, rb):
        assert x["summary"] == y["summary"]
        np.testing.assert_array_equal(x["feature"], y["feature"])
        np.testing.assert_array_equal(x["state"]["pose"], y["state"]["pose"])
        np.testing.assert_array_equal(x["prior"]["J0"], y["prior"]["J0"])
    tables.close()


def test_small_table_fed_upload_right_behind_a_large_one(be):
    """uploads of >= 32 table-fed windows pack on the copy stream, smaller ones on the main stream, and both use the tables'
    shared histogram / layout / slot scratch. A 1-window upload issued right behind a 64-window one from the same tables — no table
    operation in between that would have waited for the large batch's pack kernel — must not rewrite the scratch under it, and the other
    way round: both batches bit for bit what they are when uploaded alone, several times in a row."""
    W = 64
    base = [gf.stream.Stream(seed=41 + q, n_kf=12, new_per_frame=14 + 9 * q) for q in range(4)]
    streams = [base[w % 4] for w in range(W)]
    tables = abi.FeatureTables(be.lib, "gfbe_", be.ctx, n_tables=W, capacity=1024)
    _fill_tables(tables, streams)
    win = [_stream_window(be, S) for S in base]
    snaps = [win[w % 4][0] for w in range(W)]
    tables.triangulate([win[w % 4][1] for w in range(W)], [win[w % 4][2] for w in range(W)], with_depth=False)

    def run(b):
        b.solve(abi.MARGIN_OLD)
        r = b.download()
        b.free()
        return r

    want_big = run(be.batch_upload_tables(tables, snaps))
    want_one = run(be.batch_upload_tables(tables, snaps[:1]))
    for rep in range(4):
        big = be.batch_upload_tables(tables, snaps)            # copy stream
        one = be.batch_upload_tables(tables, snaps[:1])        # main stream, straight behind it
        big2 = be.batch_upload_tables(tables, snaps)           # and a large one straight behind the small one
        got_one, got_big, got_big2 = run(one), run(big), run(big2)
        for want, got in ((want_one, got_one), (want_big, got_big), (want_big, got_big2)):
            for x, y in zip(want, got):
                assert x["summary"] == y["summary"]
                np.testing.assert_array_equal(x["feature"], y["feature"])
                np.testing.assert_array_equal(x["state"]["pose"], y["state"]["pose"])
                np.testing.assert_array_equal(x["prior"]["J0"], y["prior"]["J0"])
    tables.close()
