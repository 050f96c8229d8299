import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Where the host-to-host time of ONE window goes on the host's clock: the upload call (scan, pack, enqueue of the ingest / expand
kernels), the solve call (enqueue of ~45 launches), the download call (wait + unpack) — against gfbe_solve_window as one call."""
import os, time, numpy as np, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=int(os.environ.get("L", "2000")), use_wheel=True)
be = gf.Backend(0)
r = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
h = abi.WindowHolder(snap)
N = 200
tw = []
for _ in range(N + 20):
    t0 = time.perf_counter(); be.solve_raw(h, abi.MARGIN_OLD); tw.append(time.perf_counter() - t0)
print("gfbe_solve_window: median %.1f us" % (np.median(tw[20:]) * 1e6))
ws = gf.WindowSet([snap])
bufs = gf.DownloadBuffers(1, h.n_feature)
tu, ts, td, tf, tt = [], [], [], [], []
for _ in range(N + 20):
    t0 = time.perf_counter()
    b = be.batch_upload(ws)
    t1 = time.perf_counter()
    b.solve(abi.MARGIN_OLD)
    t2 = time.perf_counter()
    b.download_into(bufs)
    t3 = time.perf_counter()
    b.free()
    t4 = time.perf_counter()
    tu.append(t1 - t0); ts.append(t2 - t1); td.append(t3 - t2); tf.append(t4 - t3); tt.append(t4 - t0)
m = lambda v: np.median(v[20:]) * 1e6
print("three calls: upload %.1f us, solve (enqueue) %.1f us, download (wait + unpack) %.1f us, free %.1f us; total %.1f us" % (m(tu), m(ts), m(td), m(tf), m(tt)))
one = be.batch_upload([snap])
for _ in range(10): one.solve(abi.MARGIN_OLD)
torch.cuda.synchronize()
tr, te = [], []
for _ in range(N):
    t0 = time.perf_counter(); one.solve(abi.MARGIN_OLD); t1 = time.perf_counter(); torch.cuda.synchronize(); tr.append(time.perf_counter() - t0); te.append(t1 - t0)
print("resident: solve call (enqueue) %.1f us, until the device is idle %.1f us" % (np.median(te) * 1e6, np.median(tr) * 1e6))
