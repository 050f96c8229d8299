"""The windows of the independent trust-region check (shared by tests/golden/make_golden_dogleg.py and the tests)."""
import numpy as np

from _gfbe_import import gf
from golden_util import load_case

abi, synth = gf.abi, gf.synth


def cases(orc):
    """(name, snapshot, solver keyword arguments)."""
    out = [("A", load_case("A")[0], {}), ("B", load_case("B")[0], {})]
    out.append(("cfg1", synth.Scenario(seed=20250708, n_landmarks=200, use_wheel=False).window(0), {}))
    # every optional block free + PoseSubsetParameterization masks: this window rejects steps (radius halvings, reuse)
    scn = synth.Scenario(seed=61, n_landmarks=120, use_wheel=True)
    resA = orc.solve(scn.window(0), abi.MARGIN_OLD)
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, resA["state"], 1), prior=resA["prior"])
    snap.update(ex_cam_const=0, ex_wheel_const=0, ix_wheel_const=0, td_const=0, td_wheel_const=0)
    snap["ex_cam_mask"] = np.array([0, 0, 1, 0, 0, 0], np.uint8)
    snap["ex_wheel_mask"] = np.array([0, 0, 1, 1, 1, 0], np.uint8)
    snap["ix_wheel"] = np.array([1.01, 0.99, 1.02])
    snap["td"], snap["td_wheel"] = 0.002, -0.003
    out.append(("free_masks", snap, {}))
    out.append(("retry2", load_case("B")[0], {"fail_chol_iter": 2}))
    return out
