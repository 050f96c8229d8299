import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""The phase stamps workgroup (window 0, group 0) of the Schur elimination leaves in every build (schur_body: d.timing + 8): start, first
tile's loads in, first tile staged, end — of the last launch of a single window's solve."""
import numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
first = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, first["state"], 1), prior=first["prior"])
one = be.batch_upload([snap])
for rep in range(4):
    one.solve(abi.MARGIN_NONE)
    t = one.debug_timing(0)[8:16]
    print("k_schur_visblock_small, workgroup (0, group 0) of the last launch: loads of its first tile in %.2f us after its start, staged %.2f, all %d tiles of the group done %.2f" %
          ((t[1] - t[0]) * 0.01, (t[2] - t[0]) * 0.01, int(t[4]), (t[3] - t[0]) * 0.01))
