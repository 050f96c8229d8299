"""Multi-frame synthetic stream through the back end: the steady-state loop of Estimator::processImage
(estimator.cpp:1133-1215: addFeatureCheckParallax -> triangulate -> optimization -> movingConsistencyCheckW /
removeOutlier -> slideWindow -> removeFailures) driven frame by frame, with the feature tables
and the window solve coming from ONE engine — the HIP library or the CPU oracle. Used by the trajectory-level parity
test (tests/test_gpu_stream.py); nothing here is part of the product's C ABI.

Not modelled (out of scope): the front end (tracks come from projecting a synthetic world), the
initialisation phase (the first 11 frames start from truth + noise), failure detection, re-propagation of
pre-integrations after large bias changes."""
import time

import numpy as np

from . import abi, synth


class Stream:
    """A ground robot on the SURVEY §8d arc for n_kf keyframes, a world of point landmarks with persistent ids, and the
    tracker output per keyframe (ascending ids, [x y 1 u v vx vy depth] rows)."""

    def __init__(self, seed=1, n_kf=24, new_per_frame=60, use_wheel=True, noise=True, rgbd=False):
        self.scn = synth.Scenario(seed=seed, n_landmarks=0, use_wheel=use_wheel, noise=noise, n_kf=n_kf)
        scn = self.scn
        rng = np.random.default_rng(seed + 77)
        sc = 1.0 if noise else 0.0
        self.n_kf, self.use_wheel = n_kf, use_wheel
        poses = [scn.pose(t) for t in scn.kf_t]
        frames = [dict() for _ in range(n_kf)]
        next_id = 0
        for b in range(n_kf):
            for _ in range(new_per_frame if b else 3 * new_per_frame):
                depth = rng.uniform(1.0, 10.0)
                u, v = rng.uniform(0, 640), rng.uniform(0, 480)
                pn = np.array([(u - 320) / synth.FOCAL, (v - 240) / synth.FOCAL, 1.0])
                p, R = poses[b]
                pw = R @ (scn.ric @ (pn * depth) + scn.tic) + p
                length = int(rng.integers(1, 16))
                prev = None
                for j in range(b, min(n_kf, b + length)):
                    pj, Rj = poses[j]
                    pc = scn.ric.T @ (Rj.T @ (pw - pj) - scn.tic)
                    if pc[2] < 0.5 or abs(pc[0] / pc[2]) > 320 / synth.FOCAL or abs(pc[1] / pc[2]) > 240 / synth.FOCAL:
                        break
                    xy = pc[:2] / pc[2] + sc * rng.normal(0, 0.5 / synth.FOCAL, 2)
                    vel = (xy - prev) / synth.KF_DT if prev is not None else np.zeros(2)
                    prev = xy
                    dep = pc[2] * (1 + sc * rng.normal(0, 0.01)) if rgbd else 0.0
                    frames[j][next_id] = [xy[0], xy[1], 1.0, synth.FOCAL * xy[0] + 320, synth.FOCAL * xy[1] + 240, vel[0], vel[1], dep]
                next_id += 1
        self.frames = []
        for fr in frames:
            ids = sorted(fr)
            self.frames.append((np.array(ids, np.int32), np.array([fr[i] for i in ids]).reshape(-1, 8)))

    def truth_pose(self, k):
        p, R = self.scn.pose(self.scn.kf_t[k])
        return p, R


def propagate(pose, sb, samples, first, g_norm):
    """processIMU's mid-point dead reckoning of (P, Q, V) over one interval (estimator.cpp:800-850), biases held."""
    P, V = pose[:3].copy(), sb[:3].copy()
    q = pose[3:] / np.linalg.norm(pose[3:])
    ba, bg = sb[3:6], sb[6:9]
    g = np.array([0.0, 0.0, g_norm])
    acc0, gyr0 = first[:3], first[3:]
    for row in samples:
        dt, acc1, gyr1 = row[0], row[1:4], row[4:7]
        R = synth.qrot(q)
        un_acc0 = R @ (acc0 - ba) - g
        un_gyr = 0.5 * (gyr0 + gyr1) - bg
        q = synth.qmul(q, np.concatenate([0.5 * un_gyr * dt, [1.0]]))
        q = q / np.linalg.norm(q)
        un_acc1 = synth.qrot(q) @ (acc1 - ba) - g
        un_acc = 0.5 * (un_acc0 + un_acc1)
        P = P + dt * V + 0.5 * dt * dt * un_acc
        V = V + dt * un_acc
        acc0, gyr0 = acc1, gyr1
    return np.concatenate([P, q]), np.concatenate([V, ba, bg])


def run_stream(engine, tables, stream, slide_fn, rgbd=False, use_mcc=True, device_handoff=False):
    """device_handoff (HIP library only): the solver reads its landmarks from the resident tables (gfbe_batch_upload_tables)
    instead of the host building the factor list from a table download.
    engine: abi.CApi (solve / build_visual_factors / preintegrate_*); tables: abi.FeatureTables (W = 1) of the same
    library; slide_fn(state_struct, flag): the library's slide_window_state. Returns dict(traj [n,7], flags, costs, sizes)."""
    scn = stream.scn
    W = abi.WINDOW_SIZE
    st = scn.initial_state(0)                       # frames 0..10: truth + noise (the initialisation phase is out of scope)
    imu_slots = [[scn.imu_raw[i]] for i in range(W)]
    wheel_slots = [[scn.wheel_raw[i]] for i in range(W)] if stream.use_wheel else None
    tic_ric = np.concatenate([scn.tic, scn.ric.ravel()])
    for fc in range(W):
        ids, obs = stream.frames[fc]
        tables.add_frame([fc], [ids], [obs], [0.0])
    prior = None
    out = dict(traj=[], flags=[], final_cost=[], n_features=[], n_landmarks=[], iterations=[], parallax=[])
    g_norm = engine.opt.g_norm

    def merged(slot):
        return (np.concatenate([s for s, _ in slot]), slot[0][1])

    # pre_integrations[i] persist from frame to frame in the reference (processIMU pushes the samples of the incoming interval,
    # slideWindow merges the two newest on MARGIN_SECOND_NEW, estimator.cpp:3790-3806): only intervals whose sample set changed
    # are integrated again
    cache_imu, cache_wheel = {}, {}

    def preintegrated(cache, slots, fn, lin):
        # key of an interval: the identities of its sample AND first-sample arrays plus the linearisation point (biases / intrinsics)
        # the integration ran at — re-propagation after a bias change integrates again. The cache entry holds references to the
        # arrays, so an id() cannot be reused by a new array while its entry is alive (a driver that builds its arrays per frame).
        lin_key = np.asarray(lin, dtype=np.float64).tobytes()
        keys = [(tuple((id(s), id(f)) for s, f in slot), lin_key) for slot in slots]
        todo = [i for i, key in enumerate(keys) if key not in cache]
        if todo:
            for i, row in zip(todo, fn([merged(slots[i]) for i in todo])):
                cache[keys[i]] = (row.copy(), list(slots[i]))
        for key in [key for key in cache if key not in keys]:
            del cache[key]
        return np.stack([cache[key][0] for key in keys])

    out["frame_s"] = []        # wall time of every frame of the loop (the first ones include one-time allocations)
    for k in range(W, stream.n_kf):
        t_frame = time.perf_counter()
        ids, obs = stream.frames[k]
        kf, _, par = tables.add_frame([W], [ids], [obs], [0.0])
        out["parallax"].append(float(par[0]))
        flag = abi.MARGIN_OLD if kf[0] else abi.MARGIN_SECOND_NEW
        poses = abi.pose_rows(st["pose"])
        if rgbd:
            tables.triangulate([poses], [tic_ric], with_depth=True)       # if (DEPTH) triangulateWithDepth (estimator.cpp:1143-1146)
        tables.triangulate([poses], [tic_ric], with_depth=False)
        snap = dict(st)
        snap["frame_count"] = W
        if not device_handoff:
            snap.update(engine.build_visual_factors(abi.ftab_to_feature_list(tables.download(0))))
        snap["imu"] = preintegrated(cache_imu, imu_slots, lambda iv: engine.preintegrate_imu(
            iv, scn.ba_est, scn.bg_est, (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)), np.concatenate([scn.ba_est, scn.bg_est]))
        snap["imu_frame"] = np.arange(W, dtype=np.int32)
        if wheel_slots is not None:
            snap["wheel"] = preintegrated(cache_wheel, wheel_slots, lambda iv: engine.preintegrate_wheel(
                iv, [1.0, 1.0, 1.0, 0.0], (synth.VEL_N_WHEEL, synth.GYR_N_WHEEL)), [1.0, 1.0, 1.0, 0.0])
            snap["wheel_frame"] = np.arange(W, dtype=np.int32)
        snap.update(ex_cam_const=1, ex_wheel_const=1, ix_wheel_const=1, td_const=1, td_wheel_const=1, prior=prior)
        if device_handoff:
            batch = engine.batch_upload_tables(tables, [snap])
            batch.solve(flag)
            res = batch.download()[0]
            batch.free()
        else:
            res = engine.solve(snap, flag)
        st = res["state"]
        prior = res["prior"]
        tables.set_depth([res["feature"]])
        out["traj"].append(st["pose"][W].copy())
        out["flags"].append(int(flag))
        out["final_cost"].append(res["summary"]["final_cost"])
        out["iterations"].append(res["summary"]["iterations"])
        out["n_landmarks"].append(len(res["feature"]))
        if use_mcc:
            poses = abi.pose_rows(st["pose"])
            rm = tables.check_outliers([poses], [tic_ric], 1)[0]
            tables.remove_outlier([rm])
        # ---- slideWindow (estimator.cpp:3700-3899)
        back = abi.pose_rows(st["pose"][:1])[0]
        cst = abi.state_from_snapshot(st)
        slide_fn(cst, flag)
        st = abi.state_to_dict(cst)
        if flag == abi.MARGIN_OLD:
            imu_slots = imu_slots[1:] + [[scn.imu_raw[k]]] if k < stream.n_kf - 1 else imu_slots[1:] + [[]]
            if wheel_slots is not None:
                wheel_slots = wheel_slots[1:] + ([[scn.wheel_raw[k]]] if k < stream.n_kf - 1 else [[]])
            new0 = abi.pose_rows(st["pose"][:1])[0]
            Rb, Rn = back[3:].reshape(3, 3), new0[3:].reshape(3, 3)
            marg = np.concatenate([back[:3] + Rb @ scn.tic, (Rb @ scn.ric).ravel()])
            new = np.concatenate([new0[:3] + Rn @ scn.tic, (Rn @ scn.ric).ravel()])
            tables.remove_back_shift_depth([marg], [new])
        else:
            imu_slots[W - 2] = imu_slots[W - 2] + imu_slots[W - 1]
            imu_slots[W - 1] = [scn.imu_raw[k]] if k < stream.n_kf - 1 else []
            if wheel_slots is not None:
                wheel_slots[W - 2] = wheel_slots[W - 2] + wheel_slots[W - 1]
                wheel_slots[W - 1] = [scn.wheel_raw[k]] if k < stream.n_kf - 1 else []
            tables.remove_front([W])
        tables.remove_failures()
        out["n_features"].append(int(tables.size()[0]))
        out["frame_s"].append(time.perf_counter() - t_frame)
        # ---- the next image: dead-reckon the newest frame through the incoming interval (processIMU)
        if k < stream.n_kf - 1:
            samples, first = scn.imu_raw[k]
            st["pose"][W], st["speed_bias"][W] = propagate(st["pose"][W], st["speed_bias"][W], samples, first, g_norm)
    out["traj"] = np.array(out["traj"])
    return out
