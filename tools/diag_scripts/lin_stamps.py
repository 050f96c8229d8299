import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
first = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, first["state"], 1), prior=first["prior"])
one = be.batch_upload([snap])
for rep in range(3):
    one.solve(abi.MARGIN_NONE)
    ts = one.debug_timing(1).view(np.uint64).astype(np.int64)
    ref = ts[0]
    print("k_lin_small<GFBE_LIN_STAMP_MODE> (last launch), us after workgroup 0 started: tiles done %.2f | imu %.2f | wheel %.2f | prior %.2f | plane/anchor %.2f | tile wg 0 %.2f | last tile wg %.2f" % tuple((ts[i] - ref) * 0.01 for i in (1, 2, 3, 4, 5, 6, 7)))
    print("   starts: first imu item %.2f, first wheel item %.2f (ends %.2f), prior %.2f" % tuple((ts[i] - ref) * 0.01 for i in (8, 9, 11, 10)))
    print("   (stamped launch modes 1 / 3) last workgroup arrived %.2f, accepting tail done %.2f" % tuple((ts[i] - ref) * 0.01 for i in (12, 13)))
    print("   tile 0: workgroup kq = 0 start %.2f, staged %.2f, steps done %.2f | the tile's last workgroup: merge starts %.2f, values in %.2f, ends %.2f" % tuple((ts[i] - ref) * 0.01 for i in (14, 15, 16, 17, 18, 19)))
    for nm, b0 in (("wheel item 0", 24), ("prior", 28)):
        print("   %s: start %.2f, raw factor / dx done %.2f, whitened / pass 1 done %.2f, products done %.2f" % ((nm,) + tuple((ts[b0 + i] - ref) * 0.01 for i in range(4))))
    print("   prior: chunk 0 in LDS %.2f, thread 0's row done %.2f, the rows of the chunk done (barrier) %.2f, chunk 1 in LDS %.2f" % tuple((ts[20 + i] - ref) * 0.01 for i in range(4)))
