import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Phase stamps of k_lm_step_fused for one window (a -DGFBE_LMS_STAMP=1 build: GFBE_LIB=.../libgfbe_lmsstamp.so): the tile workgroup 0 —
staging, the landmarks' back-substitution, its arrival — and the window's last workgroup: k_step, the dense candidate, the pair constants."""
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
first = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, first["state"], 1), prior=first["prior"])
one = be.batch_upload([snap])
for rep in range(4):
    one.solve(abi.MARGIN_NONE)
    ts = one.debug_timing(1).view(np.uint64).astype(np.int64)
    ref = ts[0]
    print("k_lm_step_fused (last launch), us after tile workgroup 0 started: sy / sv staged %.2f, frame steps staged %.2f, landmarks done %.2f, tail's loads requested %.2f | "
          "the last workgroup arrived %.2f, k_step done %.2f, dense candidate done %.2f, pair constants done %.2f" % tuple((ts[i] - ref) * 0.01 for i in range(1, 9)))
