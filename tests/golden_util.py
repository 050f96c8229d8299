import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_vi_wheel_48.npz")


def load_case(tag):
    z = np.load(GOLDEN)
    snap, prior = {}, {}
    for k in z.files:
        if k.startswith(tag + "_in_prior_"):
            prior[k[len(tag) + 10:]] = z[k]
        elif k.startswith(tag + "_in_"):
            v = z[k]
            snap[k[len(tag) + 4:]] = v if v.ndim else v.item()
    snap["prior"] = prior if prior else None
    ev = {k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(tag + "_ev_")}
    out = {k[len(tag) + 5:]: z[k] for k in z.files if k.startswith(tag + "_out_")}
    return snap, ev, out
