"""Diagnostic: phase breakdown of k_solve (last launch) for a single 2k-landmark window."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=5, n_landmarks=2000, use_wheel=True)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
b = be.batch_upload([snap])
b.solve(abi.MARGIN_OLD); b.solve(abi.MARGIN_OLD)
t = b.debug_timing(0)
names = ["perm+scale+reduce", "tile build", "cholesky", "backsub", "gram+store"]
for i, n in enumerate(names):
    print("%-20s %8.2f us" % (n, (t[i + 1] - t[i]) * 0.01))
print("total %.2f us" % ((t[5] - t[0]) * 0.01))
