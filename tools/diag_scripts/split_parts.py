import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Resident throughput of 8192 windows by the number of parts the batch is solved in (gfbe_options.split_batch), one box."""
import os, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
snaps = []
be0 = gf.Backend(0)
for k in range(8):
    scn = synth.Scenario(seed=20250708 + k, n_landmarks=2000, use_wheel=True)
    r = be0.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
    snaps.append(scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"]))
be0.close()
B = int(os.environ.get("B", "8192"))
for parts in [int(x) for x in os.environ.get("PARTS", "1,4,6,8,4").split(",")]:
    o = abi.default_options(); o.split_batch = parts
    be = gf.Backend(0, options=o)
    b = be.batch_upload((snaps * (B // 8 + 1))[:B])
    for _ in range(2): b.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); b.solve(abi.MARGIN_OLD); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("split_batch %d: %d resident windows %.1f ms per solve, %.1fk solves/s" % (parts, B, np.median(ts) * 1e3, B / np.median(ts) / 1e3), flush=True)
    b.free(); be.close()
