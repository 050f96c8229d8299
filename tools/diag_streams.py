import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Diagnostic: 256 resident windows solved as G groups on G contexts / streams (kernels of different groups overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=2000, use_wheel=True) for u in range(8)]
be0 = gf.Backend(0)
firsts = be0.solve_batch([s.window(0) for s in scns], abi.MARGIN_OLD)
snaps = [s.window(1, state=synth.shift_state_for_next_window(s, r["state"], 1), prior=r["prior"]) for s, r in zip(scns, firsts)]
B = int(os.environ.get('B', '1024'))
opt = abi.default_options(); opt.split_batch = 0
for G in (1, 2, 3, 4, 6, 8):
    bes = [gf.Backend(0, options=opt) for _ in range(G)]
    batches = [b.batch_upload([snaps[i % 8] for i in range(B // G)]) for b in bes]
    def step():
        for bt in batches:
            bt.solve(abi.MARGIN_OLD)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("groups %d: %.3f ms per %d solves = %.0f solves/s" % (G, dt * 1e3, B // G * G, (B // G * G) / dt))
    for bt in batches: bt.free()
