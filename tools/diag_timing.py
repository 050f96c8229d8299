import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
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
print("panel 0: panel multiply (incl. barrier) %.2f us, trailing update %.2f us; wave 0: own trailing tiles + factor-and-invert of tile (1,1) %.2f us" % ((t[18]-t[17])*0.01, (t[19]-t[18])*0.01, (t[20]-t[18])*0.01))
print("   chol_inv_tile16 of tile (1,1): factor %.2f us, inverse %.2f us" % ((t[22]-t[21])*0.01, (t[23]-t[22])*0.01))

print("k_schur WG(0,0): first prefetch %.2f us, first LDS stage %.2f us, all %d tiles %.2f us" % ((t[9]-t[8])*0.01, (t[10]-t[9])*0.01, int(t[12]), (t[11]-t[8])*0.01))
print("k_schur WG(0,0) shader clock during the kernel: %.0f MHz" % ((t[14]-t[13]) / ((t[11]-t[8])*0.01)))
an = [(6, 7, "k_visblock")]
for i0, i1, nme in an:
    print("k_assemble %-24s %8.2f us" % (nme, (t[i1] - t[i0]) * 0.01))
mn = ["setup", "assemble A,b", "15x15 eig", "Schur (T, A', b')", "tail + meta", "k_marg_ldlt (incl. launch gap)"]
for i, nme in enumerate(mn):
    print("k_marg %-20s %8.2f us" % (nme, (t[25 + i] - t[24 + i]) * 0.01))
# the same under load: 256 windows
b2 = be.batch_upload([snap] * 256)
b2.solve(abi.MARGIN_OLD); b2.solve(abi.MARGIN_OLD)
t = b2.debug_timing(0)
for i, n in enumerate(names):
    print("B=256 %-20s %8.2f us" % (n, (t[i + 1] - t[i]) * 0.01))
print("B=256 k_schur WG(0,0): all %d tiles %.2f us, shader clock %.0f MHz" % (int(t[12]), (t[11]-t[8])*0.01, (t[14]-t[13]) / ((t[11]-t[8])*0.01)))
