import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""What batches solved SIDE BY SIDE would gain over batches solved one after the other (gfbe_batch_solve puts every batch of a context on
the context's one stream): K contexts, a resident batch of B windows each, all enqueued before anything is waited for."""
import os, time, numpy as np, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
B = int(os.environ.get("B", "1024"))
scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=2000, use_wheel=True) for u in range(8)]
bes = [gf.Backend(0) for _ in range(4)]
firsts = bes[0].solve_batch([s.window(0) for s in scns], abi.MARGIN_OLD)
snaps = [s.window(1, state=synth.shift_state_for_next_window(s, r["state"], 1), prior=r["prior"]) for s, r in zip(scns, firsts)]
for K in (1, 2, 3, 4):
    bs = [bes[k].batch_upload([snaps[i % 8] for i in range(B)]) for k in range(K)]
    for _ in range(2):
        for b in bs: b.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        for b in bs: b.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%d contexts x %d windows side by side: %.0f solves/s" % (K, B, K * B * 12 / el), flush=True)
    for b in bs: b.free()
