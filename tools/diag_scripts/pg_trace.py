import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""The 5 000-pose graph (BASELINE configs[3]) alone, for a kernel trace: N solves.   N=20 rocprofv3 --kernel-trace --stats -- python tools/diag_scripts/pg_trace.py"""
import os, time
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
g = synth.pose_graph(n=int(os.environ.get("POSES", "5000")))
dev = abi.PoseGraph(be.lib, "gfbe_", be.ctx)
dev.solve(g)
ts = []
for _ in range(int(os.environ.get("N", "20"))):
    t0 = time.perf_counter(); r = dev.solve(g); ts.append(time.perf_counter() - t0)
print("pose graph: median %.3f ms per solve, iterations %d" % (np.median(ts) * 1e3, r["summary"]["iterations"]))
