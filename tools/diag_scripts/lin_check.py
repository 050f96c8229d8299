import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""The first linearisation of the device (gfbe_debug_vector: assembled H, g, Schur term E) against the oracle's (gfo_linearize),
for one window alone (small-batch kernels) and inside a batch of B windows (throughput kernels)."""
import os
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
L = int(os.environ.get("L", "300"))
B = int(os.environ.get("B", "40"))
orc = oracle_lib.load()
o = abi.default_options()
o.max_num_iterations = 1
be = gf.Backend(0, options=o)
scn = synth.Scenario(seed=11, n_landmarks=L, use_wheel=True)
snap = scn.window(0)
lin = orc.linearize(snap)
ND, NV = abi.DENSE_DIM, 73
Hp = lin["Hpl"]          # [L][73]
for name, snaps in (("single", [snap]), ("batch", [snap] * B)):
    b = be.batch_upload(snaps)
    b.solve(abi.MARGIN_NONE)
    w = len(snaps) - 1
    g = b.debug_vector(3, w)
    H = np.array([b.debug_vector(1000 + r, w) for r in range(ND)])
    E = np.array([b.debug_vector(2000 + r, w)[:NV] for r in range(NV)])
    H = np.tril(H) + np.tril(H, -1).T
    sc = np.abs(lin["H"]).max()
    dH = np.abs(H - lin["H"])
    print(name, "g rel err %.3e" % (np.abs(g - lin["g"]).max() / np.abs(lin["g"]).max()), " H rel err %.3e at %s" % (dH.max() / sc, np.unravel_index(dH.argmax(), dH.shape)))
    # block-wise: pose x pose (visual + inertial), by 6 x 6 block
    blk = np.array([[dH[6 * a:6 * a + 6, 6 * c:6 * c + 6].max() for c in range(11)] for a in range(11)]) / sc
    print(" worst 6x6 pose blocks (rel):\n", np.array2string(blk, precision=1, max_line_width=200))
    # E at mu = min_mu with Jacobi scaling: compare its structure only (the oracle does not expose it): symmetric, finite
    print(" E finite", np.isfinite(E).all(), "sym err %.2e" % np.abs(E - E.T).max(), "max %.3e" % np.abs(E).max())
    b.free()
