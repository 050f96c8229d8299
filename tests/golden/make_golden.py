#!/usr/bin/env python
"""Generates tests/golden/*.npz: seeded window snapshots (inputs) + the CPU oracle's outputs for
them. The reference holds no golden vectors for this path, so these pin the
oracle itself (regression) and give the GPU tests a fixture that does not need the oracle.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from _gfbe_import import gf  # noqa: E402
import oracle_lib  # noqa: E402

abi, synth = gf.abi, gf.synth
SNAP_KEYS = ["pose", "speed_bias", "ex_pose", "ex_pose_wheel", "ix_wheel", "td", "td_wheel", "frame_count",
             "vis_feature_index", "vis_imu_i", "vis_imu_j", "vis_pts_i", "vis_pts_j", "vis_vel_i", "vis_vel_j",
             "vis_td_i", "vis_td_j", "para_feature", "feature_const", "imu", "imu_frame", "wheel", "wheel_frame",
             "ex_cam_const", "ex_wheel_const", "ix_wheel_const", "td_const", "td_wheel_const"]


def pack(snap, prefix):
    d = {prefix + k: np.asarray(snap[k]) for k in SNAP_KEYS if k in snap}
    if snap.get("prior") is not None:
        for k, v in snap["prior"].items():
            d[prefix + "prior_" + k] = np.asarray(v)
    return d


def main():
    orc = oracle_lib.load()
    out = {}
    # case A: VI-wheel window without prior, MARGIN_OLD ; case B: the following window with A's prior, SECOND_NEW
    scn = synth.Scenario(seed=424242, n_landmarks=48, use_wheel=True)
    snapA = scn.window(0)
    resA = orc.solve(snapA, abi.MARGIN_OLD)
    snapB = scn.window(1, state=synth.shift_state_for_next_window(scn, resA["state"], 1), prior=resA["prior"])
    resB = orc.solve(snapB, abi.MARGIN_SECOND_NEW)
    for tag, snap, res in (("A_", snapA, resA), ("B_", snapB, resB)):
        out.update(pack(snap, tag + "in_"))
        ev = orc.eval_factors(snap, robustify=True)
        for k in ("vis_r", "vis_J", "imu_r", "imu_J", "wheel_r", "wheel_J", "prior_r"):
            out[tag + "ev_" + k] = ev[k]
        out[tag + "ev_cost"] = np.array(ev["cost"])
        for k, v in res["state"].items():
            out[tag + "out_" + k] = np.asarray(v)
        out[tag + "out_feature"] = res["feature"]
        out[tag + "out_cost_history"] = np.array(res["summary"]["cost_history"])
        out[tag + "out_accepted"] = np.array(res["summary"]["accepted"])
        pr = res["prior"]
        out[tag + "out_prior_A"] = pr["J0"].T @ pr["J0"]
        out[tag + "out_prior_b"] = pr["J0"].T @ pr["r0"]
        out[tag + "out_prior_block_id"] = pr["block_id"]
        out[tag + "out_prior_x0"] = pr["x0"]
    np.savez_compressed(os.path.join(HERE, "window_vi_wheel_48.npz"), **out)
    print("wrote", os.path.join(HERE, "window_vi_wheel_48.npz"), sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
