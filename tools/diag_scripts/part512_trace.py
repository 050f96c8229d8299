import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Workload for a kernel trace of ONE part of a throughput batch alone on the device: 512 resident 2k-landmark windows (bench.py's), N solves with
MARGIN_OLD — what every kernel of the main stream costs when nothing runs beside it (a bench step is sixteen such parts, four side by side)."""
import os, time
import numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=2000, use_wheel=True) for u in range(8)]
firsts = be.solve_batch([s.window(0) for s in scns], abi.MARGIN_OLD)
snaps = [s.window(1, state=synth.shift_state_for_next_window(s, r["state"], 1), prior=r["prior"]) for s, r in zip(scns, firsts)]
B = int(os.environ.get("B", "512"))
batch = be.batch_upload([snaps[i % 8] for i in range(B)])
for _ in range(2): batch.solve(abi.MARGIN_OLD)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = int(os.environ.get("N", "6"))
for _ in range(N): batch.solve(abi.MARGIN_OLD)
torch.cuda.synchronize()
print("B = %d: %.3f ms per solve call" % (B, (time.perf_counter() - t0) / N * 1e3))
