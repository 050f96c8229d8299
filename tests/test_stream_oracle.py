"""CPU run of the multi-frame stream harness (ground-fusion2_amd/stream.py) on the oracle alone: the loop
addFeatureCheckParallax -> triangulate -> optimization -> movingConsistencyCheckW -> slideWindow -> removeFailures
stays anchored to the ground truth over 18 solves with both marginalisation flavours."""
import ctypes as C

import numpy as np

from _gfbe_import import gf

abi, stream = gf.abi, gf.stream


def test_oracle_stream_tracks_ground_truth(oracle):
    S = stream.Stream(seed=3, n_kf=28, new_per_frame=50)
    T = abi.FeatureTables(oracle.lib, "gfo_", None, 1, 8192, options=dict(min_parallax=14.0 / 600))
    out = stream.run_stream(oracle, T, S, lambda st, flag: oracle.lib.gfo_slide_window_state(C.byref(st), int(flag)))
    T.close()
    assert len(out["traj"]) == 18
    assert abi.MARGIN_OLD in out["flags"] and abi.MARGIN_SECOND_NEW in out["flags"]
    err = np.array([np.linalg.norm(out["traj"][i, :3] - S.truth_pose(10 + i)[0]) for i in range(18)])
    assert err.max() < 0.05                 # initial states are truth + N(0, 2 cm): the gauge keeps that offset
    assert abs(err[-1] - err[0]) < 0.01     # and it does not drift
    assert min(out["n_landmarks"]) > 300
