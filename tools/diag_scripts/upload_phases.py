import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
os.environ["GFBE_DEBUG_UPLOAD"] = "1"
be = gf.Backend(0, so=gf.backend.DIAG_SO)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
wh = abi.WindowHolder(snap)
for _ in range(12):
    b = be.batch_upload([wh]); b.free()
