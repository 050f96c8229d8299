import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Phase time stamps of ONE wave of k_vis<0> (window B/2, tile 0, first iteration) while the whole batch runs: needs a library built
with -DGFBE_KVIS_STAMP=1 (tools/diag_variants.py variant `stamp`). Ticks are 10 ns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
opt = abi.default_options(); opt.split_batch = 0
be = gf.Backend(0, options=opt, so=os.environ.get("GFBE_LIB"))
scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=2000, use_wheel=True) for u in range(8)]
snaps = [s.window(0) for s in scns]
B = int(os.environ.get("B", "512"))
batch = be.batch_upload([snaps[i % 8] for i in range(B)])
for rep in range(3):
    batch.solve(abi.MARGIN_OLD); torch.cuda.synchronize()
    t = batch.debug_timing(B)
    names = {0: "start", 1: "poses+pair consts in LDS", 2: "landmark loads issued"}
    print("rep %d: wave lifetime %.2f us" % (rep, (t[30] - t[0]) * 0.01))
    print("  prologue: poses %.2f us, first loads %.2f us" % ((t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01))
    for k in range(5):
        b = 3 + 5 * k
        if t[b + 4] <= 0: break
        nxt = t[b + 5] if (k < 4 and t[b + 5] > 0) else t[30]
        print("  step %d: eval + panel row %.2f  wait for the next observation %.2f  row stores + LDS operands + MFMA %.2f  fold + partial stores %.2f  to the next step %.2f  (step total %.2f us)" % (
            k, (t[b + 1] - t[b]) * 0.01, (t[b + 2] - t[b + 1]) * 0.01, (t[b + 3] - t[b + 2]) * 0.01, (t[b + 4] - t[b + 3]) * 0.01, (nxt - t[b + 4]) * 0.01, (nxt - t[b]) * 0.01))
