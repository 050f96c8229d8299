import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Diagnostic: per-launch time of the first-iteration k_vis<0> launch at B=256 (ablation builds allowed to fail numerically)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
opt = abi.default_options(); opt.split_batch = int(os.environ.get('SPLIT', '0'))
be = gf.Backend(0, options=opt)
scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=int(os.environ.get("L", "2000")), use_wheel=True) for u in range(8)]
snaps = [s.window(0) for s in scns]
B = int(os.environ.get("B", "256"))
batch = be.batch_upload([snaps[i % 8] for i in range(B)])
def run():
    try:
        batch.solve(abi.MARGIN_OLD)
    except Exception as e:
        pass
run(); torch.cuda.synchronize()
be.profile_enable(True); be.profile_reset()
for _ in range(3): run()
torch.cuda.synchronize()
for p in sorted(be.profile(), key=lambda p: -p["total_ms"]):
    print("%-20s launches %4d  avg %.4f ms  total %.3f" % (p["name"], p["launches"], p["total_ms"] / max(p["launches"], 1), p["total_ms"]))
# whole-solve throughput of this build (not profiled), and a result fingerprint
import time
be.profile_enable(False)
for _ in range(2): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
print("solves_per_s %.0f" % (5 * B / (time.perf_counter() - t0)))
try:
    print("final_cost %.12g" % batch.download()[0]["summary"]["final_cost"])
except Exception as e:
    print("final_cost error", e)
