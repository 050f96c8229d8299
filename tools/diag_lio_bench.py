import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Timing of gfbe_lio_linearize (one LiDAR scan's point-to-plane factors: residuals, Jacobians, J^T J, J^T r, cost; host buffers in,
host buffers out) vs the CPU oracle on one core, median of five calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from _gfbe_import import gf
from test_lio_oracle import scan
abi = gf.abi
be, orc = gf.Backend(0), oracle_lib.load()
for ct in (0, 1):
    for n in (2000, 100000, 1000000):
        args = scan(np.random.default_rng(n), n, bool(ct))
        pts, normals, offs, alpha, w, pb, pe = args
        call = lambda lib, pre, ctx: abi.lio_linearize(lib, pre, ctx, ct, pts, normals, offs, alpha, w, 0.8, pb, pe)
        call(be.lib, "gfbe_", be.ctx)
        td, tr = [], []
        for _ in range(5):
            t0 = time.perf_counter(); call(be.lib, "gfbe_", be.ctx); td.append(time.perf_counter() - t0)
        for _ in range(3):
            t0 = time.perf_counter(); call(orc.lib, "gfo_", None); tr.append(time.perf_counter() - t0)
        print("%s factor, n = %7d: device %.3f ms, oracle (1 core) %.3f ms, ratio %.1f" % ("CT   " if ct else "plain", n, sorted(td)[2] * 1e3, sorted(tr)[1] * 1e3, sorted(tr)[1] / sorted(td)[2]))
