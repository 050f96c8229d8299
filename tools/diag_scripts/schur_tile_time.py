import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""k_schur under load: the stamps of workgroup (window 0, group 0) in a batch of B 2k-landmark windows, and the kernel's launch time."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
B = int(os.environ.get("B", "2048"))
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
b = be.batch_upload([snap] * B)
b.solve(abi.MARGIN_OLD); b.solve(abi.MARGIN_OLD)
t = b.debug_timing(0)
print("B=%d k_schur WG(0,0): first prefetch %.2f us, first LDS stage %.2f us, all %d tiles %.2f us (%.2f us per tile), shader clock %.0f MHz" %
      (B, (t[9] - t[8]) * 0.01, (t[10] - t[9]) * 0.01, int(t[12]), (t[11] - t[8]) * 0.01, (t[11] - t[8]) * 0.01 / max(t[12], 1), (t[14] - t[13]) / ((t[11] - t[8]) * 0.01)))
te = b.debug_timing(B)
print("k_visasm, window 0 (thread 0): gather partials %.2f us, S_f %.2f, T^T M T %.2f, (barrier +) stage tables %.2f, H entries %.2f, E %.2f, g %.2f; total %.2f us" % tuple([(te[24 + i + 1] - te[24 + i]) * 0.01 for i in range(7)] + [(te[31] - te[24]) * 0.01]))
be.profile_enable(True); be.profile_reset()
for _ in range(3):
    b.solve(abi.MARGIN_OLD)
torch.cuda.synchronize()
for p in be.profile():
    if p["launches"] and p["name"] in ("k_schur_iter0", "k_schur", "k_vis_lin_iter0", "k_assemble_iter0", "k_lm_step_iter0", "k_vis_cost", "k_solve_iter0", "k_dense", "k_dense_iter0", "k_dense_cost", "k_candidate", "marginalize"):
        print("  %-18s %8.1f us per launch" % (p["name"], 1e3 * p["total_ms"] / p["launches"]))
