import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""One resident 2k-landmark window solved repeatedly (the B = 1 latency path); run under rocprofv3 --kernel-trace for the per-kernel
table (profiles/summarize_rocpd.py). Prints the wall time per solve."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=int(os.environ.get("L", "2000")), use_wheel=True)
first = be.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, first["state"], 1), prior=first["prior"])
one = be.batch_upload([snap])
for _ in range(3):
    one.solve(abi.MARGIN_OLD)
torch.cuda.synchronize()
N = int(os.environ.get("N", "30"))
t = time.perf_counter()
for _ in range(N):
    one.solve(abi.MARGIN_OLD)
torch.cuda.synchronize()
print("single window resident: %.3f ms per solve" % ((time.perf_counter() - t) / N * 1e3))
r = one.download()[0]
print("iterations", r["summary"]["iterations"], "final cost %.9f" % r["summary"]["final_cost"], "device ms solve %.3f marg %.3f" % (r["perf"]["ms_solve"], r["perf"]["ms_marginalize"]))
h = abi.WindowHolder(snap)
for _ in range(3):
    be.solve_raw(h, abi.MARGIN_OLD)
t = time.perf_counter()
for _ in range(N):
    be.solve_raw(h, abi.MARGIN_OLD)
print("single window host to host (gfbe_solve_window): %.3f ms" % ((time.perf_counter() - t) / N * 1e3))
