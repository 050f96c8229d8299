import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""single-window latency on a CU-masked stream: does keeping a window's kernels on ONE XCD (one L2) shorten the dependent launches?"""
import os, sys, time, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits: words[b // 32] |= (1 << (b % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s
be = gf.Backend(0)
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
r0 = be.solve(scn.window(0), abi.MARGIN_OLD)
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
cases = {"all 256 CUs (default stream of torch)": None,
         "bits 0,8,16,..: 32 CUs": list(range(0, 256, 8)),
         "bits 0..31: 32 CUs": list(range(32)),
         "bits 0..63: 64 CUs": list(range(64)),
         "bits 0,4,8,..: 64 CUs": list(range(0, 256, 4)),
         "bits 0..127": list(range(128)),
         "all bits, masked stream": list(range(256))}
for name, bits in cases.items():
    if bits is None:
        be.set_stream(torch.cuda.current_stream().cuda_stream)
    else:
        be.set_stream(masked_stream(bits).value)
    one = be.batch_upload([snap])
    for _ in range(5): one.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t1 = time.perf_counter()
        for _ in range(20): one.solve(abi.MARGIN_OLD)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t1) / 20 * 1e3)
    res = one.download()[0]
    one.free()
    print("%-40s %.4f ms (min of 5)  final %.12e" % (name, min(ts), res["summary"]["final_cost"]), flush=True)
