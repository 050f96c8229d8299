import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys, os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo")
import torch  # noqa
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
first = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, first["state"], 1), prior=first["prior"])
one = be.batch_upload([snap])
for rep in range(3):
    one.solve(abi.MARGIN_OLD)
    ts = one.debug_timing(0)
    m = ts[24:32]
    print("k_marg phases us:", [round((m[i + 1] - m[i]) * 0.01, 1) for i in range(5)], "ldlt end - marg end:", round((m[6] - m[5]) * 0.01, 1))
if os.environ.get("LDLT"):
    ts = one.debug_timing(1)   # the block behind the windows' own slots (B = 1)
    for k in range(4):
        row = ts[k * 8:k * 8 + 6]
        print("ldlt step %d: argmax %.2f publish %.2f barrier %.2f cx+row %.2f rank1 %.2f | next step starts %.2f us later" % (
            8 + k, *[(row[i + 1] - row[i]) * 0.01 for i in range(5)], (ts[(k + 1) * 8] - row[0]) * 0.01 if k < 3 else 0))
