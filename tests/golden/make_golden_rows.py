#!/usr/bin/env python
"""Generates tests/golden/rows_f1_f3_f4.npz: small seeded inputs + the CPU oracle's outputs for the rows of SURVEY.md §8f
that were built (feature tables, global_fusion pose graph, LiDAR point-to-plane factors). Regression pins for the oracle and
oracle-free fixtures for the GPU tests. Run from the repo root:  python tests/golden/make_golden_rows.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from _gfbe_import import gf  # noqa: E402
import oracle_lib  # noqa: E402
import ftab_model as fm  # noqa: E402

abi, synth = gf.abi, gf.synth


def main():
    orc = oracle_lib.load()
    out = {}
    # ---- f3: pose graph, 60 poses
    g = synth.pose_graph(n=60, seed=11, fix_every=6)
    pg = abi.PoseGraph(orc.lib, "gfo_", None)
    ev, sol = pg.eval(g), pg.solve(g, max_iterations=5)
    for k in ("pose", "rel_i", "rel_meas", "fix_i", "fix_meas"):
        out["pg_in_" + k] = np.asarray(g[k])
    for k in ("rel_r", "rel_J", "fix_r"):
        out["pg_ev_" + k] = ev[k]
    out["pg_ev_cost"] = np.array(ev["cost"])
    out["pg_out_pose"] = sol["pose"]
    out["pg_out_cost_history"] = np.array(sol["summary"]["cost_history"])
    out["pg_out_accepted"] = np.array(sol["summary"]["accepted"])
    # ---- f4: 64 point-to-plane residuals, both factor types
    rng = np.random.default_rng(21)
    import test_lio_oracle as tl
    for ct in (0, 1):
        pts, normals, offs, alpha, w, pb, pe = tl.scan(rng, 64, bool(ct))
        if ct:
            pe = tl.plus(pb, np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.05, 3)]))
        res = abi.lio_linearize(orc.lib, "gfo_", None, ct, pts, normals, offs, alpha, w, 0.9, pb, pe)
        tag = "lio%d_" % ct
        for k, v in (("pts", pts), ("normals", normals), ("offs", offs), ("alpha", alpha if ct else np.zeros(64)), ("w", w), ("pb", pb),
                     ("pe", pe if ct else pb)):
            out[tag + "in_" + k] = np.asarray(v)
        for k in ("r", "J", "H", "g"):
            out[tag + "out_" + k] = res[k]
        out[tag + "out_cost"] = np.array(res["cost"])
    # ---- f1: a scripted sequence of feature-table operations (inputs of every step + the final table)
    rng = np.random.default_rng(31)
    T = abi.FeatureTables(orc.lib, "gfo_", None, 1, 512)
    next_id, alive = 0, []
    for step in range(15):
        ids, obs, alive, next_id = fm.random_frame(rng, next_id, alive, 10)
        fc = min(step, 10)
        out["ft_in_%02d_ids" % step] = np.array(ids, np.int32)
        out["ft_in_%02d_obs" % step] = obs
        kf, cnt, par = T.add_frame([fc], [ids], [obs], [0.001 * step])
        out["ft_out_%02d_kf_cnt" % step] = np.concatenate([kf, cnt[0]])
        if step < 10:
            continue
        L = int((T.download(0)["n_obs"] >= 4).sum())
        x = 1.0 / rng.uniform(0.5, 9.0, L)
        x[rng.random(L) < 0.1] *= -1.0
        out["ft_in_%02d_x" % step] = x
        T.set_depth([x])
        T.remove_failures()
        PR = np.concatenate([rng.normal(0, 0.1, 3), np.eye(3).ravel()])
        PN = np.concatenate([rng.normal(0, 0.1, 3), np.eye(3).ravel()])
        out["ft_in_%02d_pr" % step] = np.stack([PR, PN])
        if step % 2 == 0:
            T.remove_back_shift_depth([PR], [PN])
        else:
            T.remove_front([10])
    fin = T.download(0)
    for k, v in fin.items():
        out["ft_final_" + k] = v
    np.savez_compressed(os.path.join(HERE, "rows_f1_f3_f4.npz"), **out)
    print("wrote rows_f1_f3_f4.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
