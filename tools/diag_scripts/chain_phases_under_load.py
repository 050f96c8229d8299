import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""k_solve_chain's phases for one window alone and for windows of a batch of 512 (two workgroups per CU) — which phase pays for the neighbour?
GFBE_LIB=ground-fusion2_amd/csrc/libgfbe_diag.so python tools/diag_scripts/chain_phases_under_load.py"""
import os, numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
o = abi.default_options(); o.solve_kernel = 2
be = gf.Backend(0, options=o)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
ph = [("prologue", 0, 1), ("build", 1, 2), ("pipeline", 2, 15), ("dense chol", 15, 3), ("dense backsub", 3, 16), ("chain backsub", 16, 4), ("gram", 4, 5), ("total", 0, 5)]
for B in [int(x) for x in os.environ.get("BS", "1,256,512,1024").split(",")]:
    b = be.batch_upload([snap] * B)
    for _ in range(3): b.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    rows = []
    for w in sorted(set([0, B // 5, B // 2, (4 * B) // 5, B - 1])):
        tm = b.debug_timing(w)
        rows.append([(tm[bb] - tm[a]) * 0.01 for _, a, bb in ph])
    rows = np.array(rows)
    print("B = %4d: " % B + "  ".join("%s %.1f" % (n, v) for (n, _, _), v in zip(ph, np.median(rows, axis=0))), " (median of %d windows; total min %.1f max %.1f)" % (len(rows), rows[:, -1].min(), rows[:, -1].max()), flush=True)
    b.free()
be.close()
