#!/usr/bin/env python
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
"""Resident throughput of bench.py's default workload (8192 windows of 2000 landmarks, 8 unique) under several settings of gfbe_options on
ONE box, with the per-kernel profile of one part (2048 windows, kernels one after the other) for each:
    python tools/diag_scripts/ab_options.py merge_lin_schur=0 merge_lin_schur=1 [--batch 8192] [--steps 10] [--so path.so ...]
Every positional argument is one variant: comma-separated option=value pairs (`so=<library>` picks another build of the library)."""
import argparse
import numpy as np


def main():
    # EVERY variant in a process of its own (round 6): the first back end a process creates runs the bench workload at ~113.7k solves/s,
    # every later one — another build, a byte-identical copy of the library, even the first library again — at ~106k (measured:
    # default / copy / default in one process 113.7k / 106.2k / 105.3k; the copy alone 113.9k). A/B runs inside one process made
    # whatever came second look 7 % slower than it is.
    if "--child" not in sys.argv and len([a for a in sys.argv[1:] if not a.startswith("--") and "=" in a or a == "default"]) > 1:
        import subprocess
        flags = [a for a in sys.argv[1:] if a.startswith("--")]
        vals = {}
        i = 1
        rest = []
        while i < len(sys.argv):      # (flags with a value)
            a = sys.argv[i]
            if a in ("--batch", "--steps", "--landmarks"):
                rest += [a, sys.argv[i + 1]]; i += 2
            elif a.startswith("--"):
                rest.append(a); i += 1
            else:
                i += 1
        variants = [a for a in sys.argv[1:] if not a.startswith("--") and (a == "default" or "=" in a)]
        variants = [v for k, v in enumerate(variants) if not (k > 0 and sys.argv[sys.argv.index(v) - 1] in ("--batch", "--steps", "--landmarks"))]
        outs = []
        for v in variants:
            o = subprocess.run([sys.executable, os.path.abspath(__file__), v, "--child"] + rest, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith(v + " {")]
            print(line[0] if line else (o.stdout + o.stderr)[-500:], flush=True)
            if line:
                outs.append((v, json.loads(line[0][len(v) + 1:])))
        for v, r in outs[1:]:
            b = outs[0][1]
            print("%s vs %s: x%.4f; max rel. final-cost difference %.2e" % (v, outs[0][0], r["solves_per_s"] / b["solves_per_s"],
                  max(abs(x - y) / x for x, y in zip(b["final_cost"], r["final_cost"]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--landmarks", type=int, default=2000)
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()
    import torch
    from _gfbe_import import gf
    abi, synth = gf.abi, gf.synth
    scns = [synth.Scenario(seed=20250708 + 2 + 100 * u, n_landmarks=args.landmarks, use_wheel=True) for u in range(8)]
    snaps = None
    out = {}
    for var in args.variants:
        o = abi.default_options()
        so = None
        for kv in var.split(","):
            if not kv or kv == "default":
                continue
            k, v = kv.split("=")
            if k == "so":
                so = v
            else:
                setattr(o, k, type(getattr(o, k))(float(v)))
        be = gf.Backend(device=0, options=o, so=so) if so else gf.Backend(device=0, options=o)
        if snaps is None:
            firsts = be.solve_batch([s.window(0) for s in scns], abi.MARGIN_OLD)
            snaps = [s.window(1, state=synth.shift_state_for_next_window(s, r["state"], 1), prior=r["prior"]) for s, r in zip(scns, firsts)]
        batch = be.batch_upload([snaps[i % 8] for i in range(args.batch)])
        for _ in range(2):
            batch.solve(abi.MARGIN_OLD)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            batch.solve(abi.MARGIN_OLD)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res = batch.download(raise_on_failure=False)
        rec = {"solves_per_s": args.batch * args.steps / el, "ms_per_step": el / args.steps * 1e3,
               "final_cost": [r["summary"]["final_cost"] for r in res[:8]], "iterations": [r["summary"]["iterations"] for r in res[:8]]}
        batch.free()
        if not args.no_profile:
            part = be.batch_upload([snaps[i % 8] for i in range(min(args.batch, 2048))])
            part.solve(abi.MARGIN_OLD)
            torch.cuda.synchronize()
            be.profile_enable(True)
            for _ in range(3):
                part.solve(abi.MARGIN_OLD)
            torch.cuda.synchronize()
            prof = be.profile()
            be.profile_enable(False)
            part.free()
            tot = sum(p["total_ms"] for p in prof)
            rec["profile_us_per_launch"] = {p["name"]: round(p["total_ms"] / max(p["launches"], 1) * 1e3, 1) for p in sorted(prof, key=lambda p: -p["total_ms"]) if p["total_ms"] > 0.005 * tot or "vis_lin" in p["name"] or "linschur" in p["name"]}
            rec["profile_ms_per_solve_call"] = round(tot / 3, 3)
        be.close()
        out[var] = rec
        print(var, json.dumps(rec), flush=True)
    base = out[args.variants[0]]
    for var in args.variants[1:]:
        print("%s vs %s: x%.4f; max rel. final-cost difference %.2e" % (var, args.variants[0], out[var]["solves_per_s"] / base["solves_per_s"],
              max(abs(a - b) / a for a, b in zip(base["final_cost"], out[var]["final_cost"]))))


if __name__ == "__main__":
    main()
