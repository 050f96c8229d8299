#!/usr/bin/env python
"""Generates tests/golden/dogleg_np.npz: outputs of the INDEPENDENT numpy trust-region loop (tests/ceres_trust_region_np.py,
written from Ceres 1.14's published algorithm) on fixed windows — the golden cases A / B, BASELINE configs[0], a window with
rejected steps (every optional block free + subset masks) and one with an injected linear-solver failure. The oracle's
(tests/test_oracle_numpy.py::test_dogleg_loop_*) and, on the GPU, the HIP back end's accept / reject sequences, costs and
final radii are checked against this file.   Run from the repo root:  python tests/golden/make_golden_dogleg.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from _gfbe_import import gf  # noqa: E402
import oracle_lib  # noqa: E402
import ceres_trust_region_np as ctr  # noqa: E402
from dogleg_cases import cases  # noqa: E402


def main():
    orc = oracle_lib.load()
    out = {}
    for name, snap, kw in cases(orc):
        r = ctr.solve(orc, snap, **kw)
        for k in ("accepted", "cost_history", "radius_history", "mu_history"):
            out[name + "_" + k] = np.array(r[k])
        out[name + "_termination"] = np.array(r["termination"])
        print(name, r["accepted"], "%.6f" % r["final_cost"], r["final_radius"])
    np.savez_compressed(os.path.join(HERE, "dogleg_np.npz"), **out)


if __name__ == "__main__":
    main()
