import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Per-step time stamps of the chain pipeline (a -DGFBE_CHAIN_STAMP=1 library: GFBE_LIB=.../variants/libgfbe_chainstamp.so), one window:
when the chain wave starts / finishes each block, when wide wave 1 of each segment finishes each step — us from the pipeline's start."""
import os
import torch
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
be0 = gf.Backend(0)
r = be0.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
be0.close()
for kernel in (2, 3):
    o = abi.default_options(); o.solve_kernel = kernel
    be = gf.Backend(0, options=o)
    one = be.batch_upload([snap])
    for _ in range(5): one.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    t0, t1 = one.debug_timing(0), one.debug_timing(1)
    base = t0[2]
    raw = lambda v: (v - base) * 0.01
    # (a stamp slot this kernel's steps did not write — the two-ended chain has seven step starts, not eight — holds zero: not a time)
    us = lambda v: ("%.2f" % raw(v)) if (v > 0 and -1e3 < raw(v) < 1e5) else "n/a"
    steps = 13 if kernel == 2 else 8
    print("solve_kernel %d: pipeline %s us (build %.2f, chol %.2f)" % (kernel, us(t0[15]), (t0[2] - t0[1]) * 0.01, (t0[3] - t0[15]) * 0.01))
    print("   chain (seg 0) step starts :", " ".join(us(t1[s]) for s in range(steps)))
    print("   chain (seg 0) work done   :", " ".join(us(t1[14 + s]) for s in range(6)))
    print("   wide wave 1 (seg 0) done  :", " ".join(us(t1[20 + s]) for s in range(min(steps, 12))))
    if kernel == 3:
        print("   wide wave 1 (seg 1) done  :", " ".join(us(t1[8 + s]) for s in range(6)), " (steps 1..6)")
    one.free(); be.close()
