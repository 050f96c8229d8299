import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""End-to-end loop alone (host-fed, fresh inputs every batch) with host-side phase timers; run under rocprofv3 for the device side.
  B=512 DEPTH=3 STEPS=12 python tools/diag_e2e.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "8"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from collections import deque
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
B, DEPTH, STEPS = int(os.environ.get("B", "512")), int(os.environ.get("DEPTH", "3")), int(os.environ.get("STEPS", "12"))
opt = abi.default_options(); opt.host_threads = int(os.environ.get("THREADS", "0"))
be = gf.Backend(0, options=opt)
scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=2000, use_wheel=True) for u in range(8)]
firsts = be.solve_batch([s.window(0) for s in scns], abi.MARGIN_OLD)
snaps = [s.window(1, state=synth.shift_state_for_next_window(s, r["state"], 1), prior=r["prior"]) for s, r in zip(scns, firsts)]
sets = [gf.WindowSet([snaps[(i + q) % 8] for i in range(B)]) for q in range(2)]
bufs = gf.DownloadBuffers(B, max(h.n_feature for h in sets[0].holders))
q = deque()
for i in range(DEPTH - 1):
    b = be.batch_upload(sets[i % 2]); b.solve(abi.MARGIN_OLD); q.append(b)
T = dict(upload=0.0, solve_enqueue=0.0, download=0.0, free=0.0, pack=0.0, wait=0.0, unpack=0.0)
for s in range(STEPS + 2):
    if s == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter(); T = {k: 0.0 for k in T}
    a = time.perf_counter(); nxt = be.batch_upload(sets[(s + DEPTH - 1) % 2])
    b_ = time.perf_counter(); nxt.solve(abi.MARGIN_OLD); q.append(nxt)
    c = time.perf_counter(); cur = q.popleft(); cur.download_into(bufs)
    d_ = time.perf_counter(); h = be.host_times(); T["pack"] += 1e-3 * h["upload_pack_ms"]; T["wait"] += 1e-3 * h["download_wait_ms"]; T["unpack"] += 1e-3 * h["download_unpack_ms"]; cur.free()
    e = time.perf_counter()
    T["upload"] += b_ - a; T["solve_enqueue"] += c - b_; T["download"] += d_ - c; T["free"] += e - d_
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("e2e host-fed: %.0f solves/s, %.2f ms per batch of %d (depth %d); host ms per batch: %s" % (STEPS * B / el, 1e3 * el / STEPS, B, DEPTH,
      ", ".join("%s %.2f" % (k, 1e3 * v / STEPS) for k, v in T.items())))
while q:
    cur = q.popleft(); cur.download_into(bufs); cur.free()
