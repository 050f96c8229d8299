"""Windows with the optional in-window factors switched on (gfbe_window.use_plane / use_anchor): shared by the CPU and GPU tests."""
import numpy as np

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth


def plane_window(seed=71, L=150, anchor=True, const=0):
    """First window of a scenario with a PlaneFactor on every pose and (optionally) the PoseAnchorFactor on Pose[0]."""
    scn = synth.Scenario(seed=seed, n_landmarks=L, use_wheel=True)
    snap = scn.window(0)
    rng = np.random.default_rng(seed)
    q = synth.so3_exp(rng.normal(0, 0.01, 3))
    q[2] = 0.0
    snap["plane_R"] = q / np.linalg.norm(q)
    snap["plane_Z"] = -float(snap["ex_pose_wheel"][2]) + 0.02
    snap["plane"] = dict(noise_inv=[100.0, 100.0, 50.0], const=const)      # PITCH_N_INV, ROLL_N_INV, ZPW_N_INV
    if anchor:
        snap["anchor"] = dict(pose=snap["pose"][0].copy(), sqrt_info=120.0)
    return scn, snap


def next_plane_window(scn, snap, res):
    """The following window: shifted state, the prior of `res` (it carries the 4-wide plane_R block and plane_Z), plane still on."""
    st = synth.shift_state_for_next_window(scn, res["state"], 1)
    nxt = scn.window(1, state=st, prior=res["prior"])
    nxt["plane_R"], nxt["plane_Z"], nxt["plane"] = res["state"]["plane_R"], res["state"]["plane_Z"], snap["plane"]
    return nxt
