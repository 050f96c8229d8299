#!/usr/bin/env python
"""Test infrastructure (bench.py's cpu_baseline.all_cores leg): the CPU oracle's optimization() on every host core at once, ONE PROCESS
per core. usage: cpu_all_cores.py <pickle with {"snaps": [...], "seconds": s, "procs": n}>; prints one JSON line.
This process never touches the GPU (it only loads oracle/libgfo.so); the children are forked before any of them solves, wait for a
common wall-clock instant, solve their window (child k: window k mod len(snaps)) until the deadline and report their count through a pipe."""
import json, os, pickle, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    job = pickle.load(open(sys.argv[1], "rb"))
    import ctypes as C
    import numpy as np
    from _gfbe_import import gf
    import oracle_lib
    abi = gf.abi
    orc = oracle_lib.load()
    holders = [abi.WindowHolder(s) for s in job["snaps"]]
    fsolve = orc._fn("solve_window")
    fsolve.restype = abi.c_i
    nfeat = max(h.n_feature for h in holders)
    n, seconds = int(job["procs"]), float(job["seconds"])
    start = time.time() + 1.0 + 0.004 * n          # (the forks take a while on a large host)
    pipes = []
    for k in range(n):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            try:
                if hasattr(os, "sched_setaffinity"):
                    cores = sorted(os.sched_getaffinity(0))
                    os.sched_setaffinity(0, {cores[k % len(cores)]})
            except OSError:
                pass
            st, pr, sm = abi.State(), abi.PriorHolder(), abi.Summary()
            feat = np.zeros(nfeat)
            h = holders[k % len(holders)]
            cnt, busy = 0, 0.0
            while time.time() < start:
                time.sleep(0.0005)
            t0 = time.time()
            while time.time() < start + seconds:
                fsolve(orc.head, C.byref(h.c), int(abi.MARGIN_OLD), C.byref(st), abi._pd(feat), C.byref(pr.c), C.byref(sm))
                cnt += 1
            busy = time.time() - t0
            os.write(w, ("%d %.6f\n" % (cnt, busy)).encode())
            os._exit(0)
        os.close(w)
        pipes.append((pid, r))
    total, busy = 0, 0.0
    for pid, r in pipes:
        data = b""
        while True:
            chunk = os.read(r, 256)
            if not chunk:
                break
            data += chunk
        os.close(r)
        os.waitpid(pid, 0)
        c, b = data.decode().split()
        total += int(c); busy += float(b)
    # the window of a process ends with the solve that crosses the deadline: the rate is solves over the mean busy time
    mean_busy = busy / n
    print(json.dumps({"procs": n, "solves": total, "seconds": mean_busy, "solves_per_s": total / mean_busy, "ms_per_solve_per_process": 1e3 * busy / max(total, 1)}))


if __name__ == "__main__":
    main()
