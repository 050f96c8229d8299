#!/usr/bin/env python
"""bench.py — sliding-window solves/sec of the MI355X-native back end (BASELINE.json metric).

One *step* = one pass of the hot path over one batch of synthetic input: every resident window is
reset to its uploaded state and run through the whole Estimator::optimization() sequence
(estimator.cpp:2951-3698): <= 8 dogleg iterations, double2vector re-anchoring, MARGIN_OLD
marginalisation. Workload = BASELINE.json configs[1]: 10-keyframe VI-wheel window, 2 000 landmarks,
with a marginalisation prior (SURVEY.md §8d generator), B independent windows per GPU resident in
HBM. Multi-GPU: windows are independent units -> sharded over ranks, no data-path collective
("scaling": "weak"); the barrier + max-over-ranks timing is the only communication.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="resident windows per GPU (throughput saturates near 1024: 39k solves/s at 256, 47k at 512, 49k at 1024 and 2048)")
    ap.add_argument("--landmarks", type=int, default=2000)
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic windows (tiled to --batch)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split", type=int, default=None, help="gfbe_options.split_batch: parts a batch of >= 128 windows is solved in, side by side (library default: 2)")
    ap.add_argument("--graph", action="store_true", help="gfbe_options.use_graph: replay the launch sequence as a hipGraph")
    ap.add_argument("--shard-landmarks", action="store_true",
                    help="N > 1 only: every rank holds the SAME windows and evaluates its share of the landmark tiles; the partial "
                         "normal equations are summed with RCCL all-reduces (BASELINE configs[2]; strong scaling of one solve)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from _gfbe_import import gf
    abi, synth = gf.abi, gf.synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # "nccl" is RCCL on ROCm. GFBE_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs than
        # ranks (ranks then share devices): a test aid, never used by the driver.
        dist.init_process_group(os.environ.get("GFBE_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    opts = abi.default_options()
    if args.split is not None:
        opts.split_batch = args.split
    if args.graph:
        opts.use_graph = 1
    be = gf.Backend(device=local_rank, options=opts)          # raises if the HIP extension / GPU is missing
    be.set_stream(torch.cuda.current_stream().cuda_stream)
    shard = args.shard_landmarks and world > 1
    if shard:
        be.set_allreduce(gf.dist.torch_allreduce_hook(), rank, world)

    # ---- synthetic input (untimed). The prior of each window comes from the back end itself:
    # window k of the run is solved + marginalised (MARGIN_OLD) on the GPU, its prior and shifted
    # state seed window k+1 — exactly what consecutive optimization() calls do.
    t0 = time.time()
    scns = [synth.Scenario(seed=20250708 + 2 + 100 * u + (0 if shard else 7919 * rank), n_landmarks=args.landmarks, use_wheel=True)
            for u in range(args.unique)]
    firsts = be.solve_batch([s.window(0) for s in scns], abi.MARGIN_OLD)
    snaps = []
    for s, r in zip(scns, firsts):
        st = synth.shift_state_for_next_window(s, r["state"], 1)
        snaps.append(s.window(1, state=st, prior=r["prior"]))
    K_per = [len(s["vis_imu_i"]) for s in snaps]
    batch_snaps = [snaps[i % args.unique] for i in range(args.batch)]
    batch = be.batch_upload(batch_snaps)
    K_batch = sum(K_per[i % args.unique] for i in range(args.batch))
    setup_s = time.time() - t0

    def step():
        batch.solve(abi.MARGIN_OLD)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t_start
    # whole-job aggregate: SUM of solves over ranks / MAX of elapsed over ranks
    solves, elapsed = gf.dist.aggregate_throughput(args.batch * args.steps, elapsed, dist if world > 1 else None)
    if shard:
        solves //= world          # every rank worked on the same windows
    value = solves / elapsed

    # ---- correctness of what was timed (cheap): every window converged to the same cost as window 0 of its kind
    res = batch.download()
    final_costs = [r["summary"]["final_cost"] for r in res[: args.unique]]
    iters = [r["summary"]["iterations"] for r in res[: args.unique]]

    out = None
    prof = None
    if rank == 0 or shard:   # (landmark sharding: the all-reduces need every rank to take the same steps)
        # ---- roofline of the dominant kernel, measured live with hipEvents on the launch stream
        be.profile_enable(True)
        be.profile_reset()
        nprof = 3
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        prof = {p["name"]: p for p in be.profile()}
        be.profile_enable(False)
    if rank == 0:
        tot_ms = sum(p["total_ms"] for p in prof.values())
        dom = max((p for p in prof.values()), key=lambda p: p["total_ms"])
        lin0 = prof.get("k_vis_lin_iter0", prof.get("k_vis_lin"))
        # Dominant hot-path kernel: k_vis<0> = visual evaluate + linearise with J^T J fused on the FP64
        # matrix cores (J is never materialised). SURVEY.md §8d per-unit figures for one visual residual
        # block: 108 B of input (fused form) and 1.6 kflop of J^T J. At 14.8 flop/B the kernel sits to the
        # right of the FP64 ridge point (78.6 TF / 8 TB/s = 9.8 flop/B): the matrix-core roofline bounds it.
        lin_ms = lin0["total_ms"] / max(lin0["launches"], 1)
        # (batches >= 128 windows run as two halves: a launch then covers half of the resident windows)
        units_per_launch = K_batch * nprof / max(lin0["launches"], 1)
        algo_flops = 1600.0 * units_per_launch
        algo_bytes = 108.0 * units_per_launch
        achieved_tf = algo_flops / (lin_ms * 1e-3) / 1e12
        # HBM traffic of that kernel: PMC counters cannot be read from inside this process; the committed rocprofv3
        # passes (profiles/r1_pmc_fetch.txt / _write.txt, tests/diag_pmc.sh) hold FETCH_SIZE / WRITE_SIZE per dispatch
        # for exactly the default workload, so they are quoted for it and left null for any other configuration.
        traffic = None
        if args.batch == 1024 and args.landmarks == 2000 and args.unique == 8:
            try:
                tot = 0.0
                for fn, key in (("r1_pmc_fetch.txt", "FETCH_SIZE"), ("r1_pmc_write.txt", "WRITE_SIZE")):
                    lines = open(os.path.join(ROOT, "profiles", fn)).read().splitlines()
                    i = [k for k, ln in enumerate(lines) if "k_visILi0E" in ln and "grid=(32768,36,1)" in ln][0]   # one half of the batch
                    tot += float([ln for ln in lines[i + 1:i + 4] if key in ln][0].split()[1]) * 1024.0
                traffic = tot
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": "k_vis<0> (visual evaluate + linearise + fused J^T J; first iteration: all windows active)",
                    "achieved": achieved_tf, "peak": 78.6, "unit": "TFLOP/s", "frac": achieved_tf / 78.6, "traffic": traffic,
                    "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the same workload, bytes per launch: profiles/r1_pmc_fetch.txt + r1_pmc_write.txt" if traffic else None,
                    "avg_launch_ms": lin_ms, "algorithmic_flops_per_launch": algo_flops,
                    "algorithmic_bytes_per_launch": algo_bytes, "hbm_view_GBps": algo_bytes / (lin_ms * 1e-3) / 1e9,
                    "hbm_view_frac_of_8TBps": algo_bytes / (lin_ms * 1e-3) / 8e12,
                    "dominant_by_time": dom["name"],
                    "time_share": {k: round(v["total_ms"] / tot_ms, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}}

        # ---- single-window latency (B = 1), same workload
        single_ms = None
        if not shard:
            one = be.batch_upload(batch_snaps[:1])
            for _ in range(3):
                one.solve(abi.MARGIN_OLD)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                one.solve(abi.MARGIN_OLD)
            torch.cuda.synchronize()
            single_ms = (time.perf_counter() - t1) / 10 * 1e3
            one.free()

        # ---- CPU baseline: the oracle (Ceres stand-in "port", 1 core) on the same windows, bounded sample
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.load()
            n_done, t_cpu = 0, 0.0
            holders = [abi.WindowHolder(s) for s in snaps]
            while t_cpu < args.cpu_seconds:
                h = holders[n_done % len(holders)]
                tc = time.perf_counter()
                r = orc.solve(h, abi.MARGIN_OLD)
                t_cpu += time.perf_counter() - tc
                if n_done < len(holders):
                    ref_cost = r["summary"]["final_cost"]
                    assert abs(final_costs[n_done] - ref_cost) < 1e-6 * ref_cost, (final_costs[n_done], ref_cost)
                n_done += 1
            cpu = {"value": n_done / t_cpu, "unit": "solves/s", "cores": 1, "kind": "port",
                   "sample": "%d full optimization() calls (solve + MARGIN_OLD) of the same %d-landmark windows in %.1f s; "
                             "oracle/ C++ restatement, -O3 -march=native, 1 thread like the reference's ceres::Solve" %
                             (n_done, args.landmarks, t_cpu),
                   "ms_per_solve": 1e3 * t_cpu / n_done}
            # the same port on every host core (one window per thread; ctypes drops the GIL): SURVEY.md §8d (b)
            import threading
            ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            counts = [0] * ncore
            deadline = time.perf_counter() + min(args.cpu_seconds, 10.0)
            import ctypes as C
            fsolve = orc._fn("solve_window")
            fsolve.restype = abi.c_i
            def worker(k):   # the bare C call in the loop: no Python-side result conversion under the GIL
                hs = [abi.WindowHolder(s) for s in snaps]
                st, pr, sm = abi.State(), abi.PriorHolder(), abi.Summary()
                feat = np.zeros(max(h.n_feature for h in hs))
                i = k
                while time.perf_counter() < deadline:
                    fsolve(orc.head, C.byref(hs[i % len(hs)].c), int(abi.MARGIN_OLD), C.byref(st), abi._pd(feat), C.byref(pr.c), C.byref(sm))
                    counts[k] += 1
                    i += 1
            t_all = time.perf_counter()
            th = [threading.Thread(target=worker, args=(k,)) for k in range(ncore)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            t_all = time.perf_counter() - t_all
            cpu["all_cores"] = {"value": sum(counts) / t_all, "unit": "solves/s", "cores": ncore,
                                "sample": "%d solves on %d threads in %.1f s" % (sum(counts), ncore, t_all)}
        out = {
            "metric": "sliding-window solves/sec (10-kf, 2k landmarks)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 10-keyframe VI-wheel window, %d landmarks (K=%d visual factors avg), "
                                   "marginalisation prior n=%d, full optimization() = <=8 dogleg iterations + re-anchor + MARGIN_OLD" %
                                   (args.landmarks, int(np.mean(K_per)), snaps[0]["prior"]["n"]),
                       "windows_per_gpu": args.batch, "unique_windows": args.unique, "parallelism": ("landmark tiles of every window sharded over %d ranks, RCCL all-reduce of the partial normal equations" % world) if shard
                                      else "windows sharded over %d rank(s), no collective" % world},
            "roofline": roofline, "cpu_baseline": cpu,
            "single_window_ms": single_ms, "single_window_solves_per_s": (1e3 / single_ms) if single_ms else None,
            "iterations": iters, "final_cost": final_costs, "setup_s": setup_s,
        }
        if cpu:
            out["speedup_vs_cpu_1core"] = value / cpu["value"]
            if single_ms:
                out["single_window_speedup_vs_cpu_1core"] = (1e3 / single_ms) / cpu["value"]
    batch.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
