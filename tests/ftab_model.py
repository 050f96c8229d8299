"""Independent pure-Python model of the FeatureManager list operations (feature_manager.cpp:57-116, 249-302,
801-934) used to pin the C++ oracle (tests/test_ftab_oracle.py) and to drive random operation sequences. Plain
Python lists stand for std::list<FeaturePerId> / std::vector<FeaturePerFrame>; only the integer structure and the
depth hand-over of removeBackShiftDepth are modelled (triangulation is checked against geometry instead)."""
import numpy as np

WINDOW_SIZE = 10
INIT_DEPTH = 5.0


class Feat:
    def __init__(self, fid, start):
        self.id, self.start, self.obs, self.td = fid, start, [], []
        self.depth, self.eflag, self.sflag = -1.0, 0, 0


class PyFeatureManager:
    def __init__(self):
        self.feature = []

    def add_frame(self, frame_count, ids, obs8, td):
        last_track = new = long_track = 0
        for fid, row in zip(ids, obs8):
            hit = next((f for f in self.feature if f.id == fid), None)
            if hit is None:
                f = Feat(int(fid), frame_count)
                f.obs.append(np.array(row, float))
                f.td.append(td)
                self.feature.append(f)
                new += 1
            else:
                hit.obs.append(np.array(row, float))
                hit.td.append(td)
                last_track += 1
                if len(hit.obs) >= 4:
                    long_track += 1
        if frame_count < 2 or last_track < 20 or long_track < 40 or new > 0.5 * last_track:
            return True, (last_track, new, long_track), 0.0
        s, n = 0.0, 0
        for f in self.feature:
            if f.start <= frame_count - 2 and f.start + len(f.obs) - 1 >= frame_count - 1:
                oi, oj = f.obs[frame_count - 2 - f.start], f.obs[frame_count - 1 - f.start]
                s += np.hypot(oi[0] / oi[2] - oj[0], oi[1] / oi[2] - oj[1])
                n += 1
        if n == 0:
            return True, (last_track, new, long_track), 0.0
        return bool(s / n >= 10.0 / 600.0), (last_track, new, long_track), s / n * 600.0

    def remove_back_shift_depth(self, marg_R, marg_P, new_R, new_P):
        keep = []
        for f in self.feature:
            if f.start != 0:
                f.start -= 1
                keep.append(f)
                continue
            uv = f.obs[0][:3].copy()
            del f.obs[0], f.td[0]
            if len(f.obs) < 2:
                continue
            pj = new_R.T @ (marg_R @ (uv * f.depth) + marg_P - new_P)
            f.depth = pj[2] if pj[2] > 0 else INIT_DEPTH
            keep.append(f)
        self.feature = keep

    def remove_back(self):
        keep = []
        for f in self.feature:
            if f.start != 0:
                f.start -= 1
            else:
                del f.obs[0], f.td[0]
                if not f.obs:
                    continue
            keep.append(f)
        self.feature = keep

    def remove_front(self, frame_count):
        keep = []
        for f in self.feature:
            if f.start == frame_count:
                f.start -= 1
            elif f.start + len(f.obs) - 1 >= frame_count - 1:
                j = WINDOW_SIZE - 1 - f.start
                del f.obs[j], f.td[j]
                if not f.obs:
                    continue
            keep.append(f)
        self.feature = keep

    def remove_outlier(self, ids):
        s = set(int(i) for i in ids)
        self.feature = [f for f in self.feature if f.id not in s]

    def remove_failures(self):
        self.feature = [f for f in self.feature if f.sflag != 2]

    def clear_depth(self):
        for f in self.feature:
            f.depth = -1.0

    def set_depth(self, x):
        k = -1
        for f in self.feature:
            if len(f.obs) < 4:
                continue
            k += 1
            f.depth = 1.0 / x[k]
            f.sflag = 2 if f.depth < 0 else 1

    def depth_vector(self):
        return np.array([1.0 / f.depth for f in self.feature if len(f.obs) >= 4])

    def snapshot(self):
        n = len(self.feature)
        out = dict(feature_id=np.array([f.id for f in self.feature], np.int32), start_frame=np.array([f.start for f in self.feature], np.int32),
                   n_obs=np.array([len(f.obs) for f in self.feature], np.int32), obs8=np.zeros((n, 11, 8)), obs_td=np.zeros((n, 11)),
                   estimated_depth=np.array([f.depth for f in self.feature], float),
                   estimate_flag=np.array([f.eflag for f in self.feature], np.int32), solve_flag=np.array([f.sflag for f in self.feature], np.int32))
        for k, f in enumerate(self.feature):
            for o, (row, td) in enumerate(zip(f.obs, f.td)):
                out["obs8"][k, o] = row
                out["obs_td"][k, o] = td
        return out


def random_frame(rng, next_id, alive, n_new, p_lost=0.15):
    """One frame of a synthetic tracker: every live id survives with probability 1 - p_lost, n_new ids are born.
    Returns (ids ascending, obs8 rows, updated alive list, next_id)."""
    alive = [i for i in alive if rng.random() > p_lost]
    born = list(range(next_id, next_id + n_new))
    ids = sorted(alive + born)
    obs = np.column_stack([rng.normal(0, 0.3, (len(ids), 2)), np.ones(len(ids)), rng.uniform(0, 640, len(ids)), rng.uniform(0, 480, len(ids)),
                           rng.normal(0, 0.1, (len(ids), 2)), rng.uniform(0.5, 6.0, len(ids))])
    return ids, obs, alive + born, next_id + n_new


def drive(rng, tables, n_steps=40, n_new=25):
    """Apply the same random sequence of operations to every object in `tables` (PyFeatureManager and/or
    single-table FeatureTables adaptors exposing the same method names)."""
    next_id, alive, frame_count = 0, [], 0
    log = []
    for step in range(n_steps):
        ids, obs, alive, next_id = random_frame(rng, next_id, alive, int(rng.integers(5, n_new)))
        td = float(rng.normal(0, 1e-3))
        log.append([t.add_frame(frame_count, ids, obs, td) for t in tables])
        if frame_count < WINDOW_SIZE:
            frame_count += 1
            continue
        # a "solve": new depths for the landmarks with >= 4 observations, a few of them negative
        L = sum(1 for n in tables[0].snapshot()["n_obs"] if n >= 4)
        x = 1.0 / rng.uniform(0.5, 9.0, L)
        x[rng.random(L) < 0.05] *= -1.0
        for t in tables:
            t.set_depth(x)
            t.remove_failures()
        if step % 7 == 3:
            snap = tables[0].snapshot()
            drop = snap["feature_id"][rng.random(len(snap["feature_id"])) < 0.1]
            for t in tables:
                t.remove_outlier(drop)
            alive = [i for i in alive if i not in set(int(d) for d in drop)]
        choice = rng.random()
        if choice < 0.55:
            a = rng.normal(0, 0.05, 3)
            Rm = np.eye(3) + np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            Rm, _ = np.linalg.qr(Rm)
            Pm, Pn = rng.normal(0, 0.1, 3), rng.normal(0, 0.1, 3)
            for t in tables:
                t.remove_back_shift_depth(Rm, Pm, np.eye(3), Pn)
        elif choice < 0.7:
            for t in tables:
                t.remove_back()
        else:
            for t in tables:
                t.remove_front(frame_count)
        if step % 11 == 5:
            for t in tables:
                t.clear_depth()
        # ids whose track was cut by the window operations stay "alive" for the tracker: that is what happens in the
        # reference too (the front end keeps tracking an id the back end has dropped -> it is re-created)
    return log


class OneTable:
    """Adaptor: a W = 1 FeatureTables object with PyFeatureManager's method names."""

    def __init__(self, ft):
        self.ft = ft

    def add_frame(self, fc, ids, obs, td):
        kf, cnt, avg = self.ft.add_frame([fc], [ids], [obs], [td])
        return bool(kf[0]), tuple(int(c) for c in cnt[0]), float(avg[0])

    def remove_back_shift_depth(self, Rm, Pm, Rn, Pn):
        self.ft.remove_back_shift_depth([np.concatenate([Pm, Rm.ravel()])], [np.concatenate([Pn, Rn.ravel()])])

    def remove_back(self):
        self.ft.remove_back()

    def remove_front(self, fc):
        self.ft.remove_front([fc])

    def remove_outlier(self, ids):
        self.ft.remove_outlier([ids])

    def remove_failures(self):
        self.ft.remove_failures()

    def clear_depth(self):
        self.ft.clear_depth()

    def set_depth(self, x):
        self.ft.set_depth([x])

    def depth_vector(self):
        return self.ft.get_depth_vector()[0]

    def snapshot(self):
        return self.ft.download(0)


def assert_same_tables(a, b, depth_rtol=0.0):
    for k in ("feature_id", "start_frame", "n_obs", "estimate_flag", "solve_flag"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)      # integers: bit-exact
    np.testing.assert_array_equal(a["obs8"], b["obs8"])           # observations are moved, never recomputed
    np.testing.assert_array_equal(a["obs_td"], b["obs_td"])
    if depth_rtol == 0.0:
        np.testing.assert_array_equal(a["estimated_depth"], b["estimated_depth"])
    else:
        np.testing.assert_allclose(a["estimated_depth"], b["estimated_depth"], rtol=depth_rtol, atol=0)
