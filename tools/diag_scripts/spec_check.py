import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]
"""Speculative linearisation (gfbe_options.speculative_linearization, round 5) on against off, same box: every output of a solve
bit for bit on a set of windows that covers the code paths (with / without prior, free extrinsic + td, plane + anchor factors, a
filling window, rejected steps, the mu retry of a failed factorisation), then the single-window times."""
import os, time, hashlib
import torch
import numpy as np
from _gfbe_import import gf
abi, synth = gf.abi, gf.synth


def backend(spec, **kw):
    o = abi.default_options()
    o.speculative_linearization = spec
    for k, v in kw.items(): setattr(o, k, v)
    return gf.Backend(0, options=o)


def eq(x, y):
    if isinstance(x, dict): return set(x) == set(y) and all(eq(x[k], y[k]) for k in x)
    if isinstance(x, (list, tuple)) and not (x and isinstance(x[0], (int, float))): return len(x) == len(y) and all(eq(u, v) for u, v in zip(x, y))
    return np.array_equal(np.asarray(x), np.asarray(y))


def same(a, b):
    bad = []
    for k in a["state"].keys():
        if not eq(a["state"][k], b["state"][k]): bad.append("state." + k)
    if not np.array_equal(a["feature"], b["feature"]): bad.append("feature")
    if a["summary"] != b["summary"]: bad.append("summary")
    if (a.get("prior") is None) != (b.get("prior") is None): bad.append("prior?")
    elif a.get("prior") is not None:
        for k in a["prior"]:
            if not eq(a["prior"][k], b["prior"][k]): bad.append("prior." + k)
    return bad


def cases():
    out = []
    be = backend(0)
    for seed in range(6):
        scn = synth.Scenario(seed=4100 + seed, n_landmarks=[2000, 600, 200, 3000, 1200, 64][seed], use_wheel=True)
        w0 = scn.window(0)
        out.append(("first/%d" % seed, w0, {}))
        r = be.solve_batch([w0], abi.MARGIN_OLD)[0]
        w1 = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
        out.append(("prior/%d" % seed, w1, {}))
        if seed < 2:
            from test_gpu_branches import all_free
            out.append(("all_free/%d" % seed, all_free(w1), {}))
        if seed == 0:
            out.append(("mu_retry/%d" % seed, w1, dict(test_fail_chol_iter=2)))
            out.append(("mu_retry0/%d" % seed, w1, dict(test_fail_chol_iter=1, test_fail_chol_count=3)))
            out.append(("iters3/%d" % seed, w1, dict(max_num_iterations=3)))
            out.append(("iters1/%d" % seed, w1, dict(max_num_iterations=1)))
            out.append(("kernel1/%d" % seed, w1, dict(solve_kernel=1)))
            out.append(("kernel2/%d" % seed, w1, dict(solve_kernel=2)))
    be.close()
    try:
        from plane_cases import plane_window, next_plane_window
        scn, snap = plane_window(anchor=True)
        out.append(("plane", snap, {}))
        bp = backend(0)
        out.append(("plane/next", next_plane_window(scn, snap, bp.solve(snap, abi.MARGIN_OLD)), {}))
        bp.close()
    except Exception as e:
        print("(no plane case: %r)" % (e,))
    return out


bad_total = 0
for name, snap, kw in cases():
    res = []
    for spec in (0, 1):
        be = backend(spec, **kw)
        res.append((be.solve(snap, abi.MARGIN_OLD), be.solve_batch([snap] * 3, abi.MARGIN_OLD)[2], be.solve(snap, abi.MARGIN_NONE)))
        be.close()
    bad = [same(a, b) for a, b in zip(res[0], res[1])]
    s = res[1][0]["summary"]
    acc = "".join(str(int(x)) for x in s["accepted"][1:])
    h = hashlib.md5()
    for r in res[1]:
        h.update(repr(sorted((k, np.asarray(v).tolist() if not isinstance(v, dict) else repr(v)) for k, v in r["state"].items())).encode())
        h.update(np.ascontiguousarray(r["feature"]).tobytes()); h.update(np.asarray(r["summary"]["cost_history"]).tobytes())
        if r.get("prior") is not None:
            h.update(repr(sorted((k, np.asarray(v).tolist() if not isinstance(v, dict) else repr(v)) for k, v in r["prior"].items())).encode())
    print("%-14s iterations %d accepted %s termination %d  %s  digest %s" % (name, s["iterations"], acc, s["termination"], "identical" if not any(bad) else "DIFFERENT " + repr(bad), h.hexdigest()[:12]), flush=True)
    if any(bad):
        bad_total += 1
        a, b = res[0][0], res[1][0]
        print("    cost history off", [float("%.15g" % x) for x in a["summary"]["cost_history"]], a["summary"]["accepted"])
        print("    cost history on ", [float("%.15g" % x) for x in b["summary"]["cost_history"]], b["summary"]["accepted"])
        print("    dpose %.3e" % np.abs(np.asarray(a["state"]["pose"]) - np.asarray(b["state"]["pose"])).max())
print("windows with differences:", bad_total)

# ---- batches of 32 windows and more (the throughput kernel set): every window of the set above in one batch
allc = cases()
for label, pick in (("batch/plain", lambda n, kw: not kw and not n.startswith(("all_free", "plane"))), ("batch/with_free_extrinsic", lambda n, kw: not kw and not n.startswith("plane"))):
    snaps = [sn for n, sn, kw in allc if pick(n, kw)]
    snaps = (snaps * (40 // len(snaps) + 1))[:40]
    res = []
    for spec in (0, 1):
        be = backend(spec)
        res.append(be.solve_batch(snaps, abi.MARGIN_OLD))
        be.close()
    bad = [same(a, b) for a, b in zip(res[0], res[1])]
    h = hashlib.md5()
    for r in res[1]:
        h.update(repr(sorted((k, np.asarray(v).tolist() if not isinstance(v, dict) else repr(v)) for k, v in r["state"].items())).encode())
        h.update(np.ascontiguousarray(r["feature"]).tobytes()); h.update(np.asarray(r["summary"]["cost_history"]).tobytes())
        if r.get("prior") is not None:
            h.update(repr(sorted((k, np.asarray(v).tolist() if not isinstance(v, dict) else repr(v)) for k, v in r["prior"].items())).encode())
    nrej = sum(1 for r in res[1] for x in r["summary"]["accepted"][1:] if not x)
    print("%-26s %d windows, %d rejected steps  %s  digest %s" % (label, len(snaps), nrej, "identical" if not any(bad) else "DIFFERENT in %d windows %r" % (sum(1 for x in bad if x), [x for x in bad if x][:2]), h.hexdigest()[:12]), flush=True)
    if any(bad): bad_total += 1
print("windows / batches with differences:", bad_total)
if os.environ.get("NOTIMES"): raise SystemExit(0)
# ---- times
scn = synth.Scenario(seed=20250708 + 2, n_landmarks=2000, use_wheel=True)
be0 = backend(0)
r = be0.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"])
be0.close()
for spec in (0, 1, 0, 1):
    be = backend(spec)
    one = be.batch_upload([snap])
    for _ in range(20): one.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); one.solve(abi.MARGIN_OLD); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    h = abi.WindowHolder(snap)
    th = []
    for _ in range(220):
        t0 = time.perf_counter(); be.solve_raw(h, abi.MARGIN_OLD); th.append(time.perf_counter() - t0)
    one.free()
    print("speculative %d: one window resident %.4f ms (p10 %.4f)  host to host %.4f ms (p10 %.4f)" %
          (spec, np.median(ts) * 1e3, np.percentile(ts, 10) * 1e3, np.median(th[20:]) * 1e3, np.percentile(th[20:], 10) * 1e3), flush=True)
    be.close()

# ---- throughput: 8 unique 2k-landmark windows x 1024, resident
snaps = []
be0 = backend(0)
for k in range(8):
    scn = synth.Scenario(seed=20250708 + k, n_landmarks=2000, use_wheel=True)
    r = be0.solve_batch([scn.window(0)], abi.MARGIN_OLD)[0]
    snaps.append(scn.window(1, state=synth.shift_state_for_next_window(scn, r["state"], 1), prior=r["prior"]))
be0.close()
B = int(os.environ.get("B", "8192"))
for spec in (0, 1, 0, 1):
    be = backend(spec)
    b = be.batch_upload((snaps * (B // 8 + 1))[:B])
    for _ in range(2): b.solve(abi.MARGIN_OLD)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); b.solve(abi.MARGIN_OLD); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("speculative %d: %d resident windows %.1f ms per solve, %.1fk solves/s" % (spec, B, np.median(ts) * 1e3, B / np.median(ts) / 1e3), flush=True)
    b.free(); be.close()
