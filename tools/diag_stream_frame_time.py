import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Wall time per camera frame of the whole back-end loop of processImage (feature tables -> triangulate -> optimization() with
MARGIN_OLD / MARGIN_SECOND_NEW -> setDepth -> movingConsistencyCheckW -> slideWindow -> removeFailures) for ONE robot, on the
device (tables resident, landmarks handed over on the device) and on the CPU oracle (1 core). Python driver overhead included.
Output goes to profiles/r1_stream_frame_time.txt (see profiles/README.md)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib
from _gfbe_import import gf
abi, stream = gf.abi, gf.stream
NEW = int(os.environ.get("NEW", "450"))
S = stream.Stream(seed=3, n_kf=36, new_per_frame=NEW)
opts = dict(min_parallax=14.0 / 600, depth_threshold=6.0)
be = gf.Backend(device=0)
orc = oracle_lib.load()
res = {}
legs = [("device", be, be.lib, "gfbe_", True), ("device (host factor list)", be, be.lib, "gfbe_", False)]
if os.environ.get("ORACLE"):      # the oracle's dense Jacobi eigen-decomposition of Amm (the reference's O(m^3) construction) takes seconds per frame
    legs.append(("oracle, 1 core", orc, orc.lib, "gfo_", False))
for name, eng, lib, pre, handoff in legs:
    T = abi.FeatureTables(lib, pre, be.ctx if eng is be else None, 1, 16384, options=opts)
    slide = getattr(lib, pre + "slide_window_state")
    t0 = time.perf_counter()
    out = stream.run_stream(eng, T, S, lambda st, flag: slide(C.byref(st), int(flag)), device_handoff=handoff)
    dt = time.perf_counter() - t0
    n = len(out["traj"])
    res[name] = (dt / n, out)
    T.close()
o = res["device"][1]
print("one robot, %d frames after the first window; per frame: %d features in the table, %d landmarks in the window (mean), %d iterations (mean), %d x MARGIN_OLD"
      % (len(o["traj"]), np.mean(o["n_features"]), np.mean(o["n_landmarks"]), np.mean(o["iterations"]), sum(f == abi.MARGIN_OLD for f in o["flags"])))
for k, (ms, out) in res.items():
    fr = np.array(out["frame_s"]) * 1e3      # (the dead-reckoning between two frames, Python, is outside: processIMU's work)
    print("%-28s %8.2f ms per frame (whole run incl. the first window's set-up); loop body: median %.2f, mean after 3 frames %.2f, max %.2f"
          % (k, ms * 1e3, np.median(fr), fr[3:].mean(), fr.max()))
if "oracle, 1 core" in res:
    a, b = res["device"][1]["traj"], res["oracle, 1 core"][1]["traj"]
    print("trajectory device vs oracle: max |dp| = %.2e m" % np.abs(a[:, :3] - b[:, :3]).max())
a, b = res["device"][1]["traj"], res["device (host factor list)"][1]["traj"]
print("device hand-over vs host factor list: identical trajectory:", bool(np.array_equal(a, b)))

# ---- where the frame time goes (device hand-over leg): wall time per wrapped call
if os.environ.get("BREAKDOWN", "1") == "1":
    import collections
    acc = collections.defaultdict(float)
    def wrap(obj, name, label=None):
        f = getattr(obj, name)
        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            acc[label or name] += time.perf_counter() - t0
            return r
        setattr(obj, name, g)
    T = abi.FeatureTables(be.lib, "gfbe_", be.ctx, 1, 16384, options=opts)
    for m in ("add_frame", "triangulate", "set_depth", "check_outliers", "remove_outlier", "remove_back_shift_depth", "remove_front", "remove_failures", "size"):
        wrap(T, m, "ftab." + m)
    for m in ("preintegrate_imu", "preintegrate_wheel", "batch_upload_tables"):
        wrap(be, m)
    Bt = gf.backend.Batch
    for m in ("solve", "download", "free"):
        wrap(Bt, m, "batch." + m)
    slide = be.lib.gfbe_slide_window_state
    t0 = time.perf_counter()
    out = stream.run_stream(be, T, S, lambda st, flag: slide(C.byref(st), int(flag)), device_handoff=True)
    tot = time.perf_counter() - t0
    n = len(out["traj"])
    print("breakdown (ms per frame, total %.2f):" % (tot / n * 1e3))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("  %-32s %7.3f" % (k, v / n * 1e3))
    print("  %-32s %7.3f" % ("python driver / other", (tot - sum(acc.values())) / n * 1e3))
