"""Worker of tests/test_gpu_sharded.py: one rank of a landmark-sharded solve (all ranks share GPU 0; gloo)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, L, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    fail_iter = int(sys.argv[6]) if len(sys.argv) > 6 else 0      # gfbe_options.test_fail_chol_iter (test hook of the mu retry)
    import torch
    import torch.distributed as dist
    from _gfbe_import import gf
    abi, synth = gf.abi, gf.synth
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    gnss = len(sys.argv) > 7 and sys.argv[7] == "gnss"
    plain = gf.Backend(device=0)
    if gnss:      # a window with GNSS inside the solve (the 246-dim layout, k_solve_big, the GNSS factors added by rank 0) and its prior
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import gnss_window_cases as gw
        scn, tru, snap0 = gw.gnss_window(seed=20250710 + L, L=L, n_per_frame=6)
        first = plain.solve(snap0, abi.MARGIN_OLD)
        snap = gw.next_gnss_window(scn, tru, first, seed=20250710 + L)
    else:
        scn = synth.Scenario(seed=20250710 + L, n_landmarks=L, use_wheel=True)
        # window 0 (no prior) solved + marginalised unsharded -> prior and shifted state of window 1 (same on every rank)
        first = plain.solve(scn.window(0), abi.MARGIN_OLD)
        snap = scn.window(1, state=synth.shift_state_for_next_window(scn, first["state"], 1), prior=first["prior"])
    opt = abi.default_options()
    opt.test_fail_chol_iter = fail_iter
    if len(sys.argv) > 7 and sys.argv[7] == "retry3":      # three consecutive failures of that iteration's factorisation: mu x 1000, the whole ladder allowed
        opt.test_fail_chol_count = 3
        opt.sharded_mu_retries = 8
    failing = gf.Backend(device=0, options=opt)
    ref = failing.solve(snap, abi.MARGIN_OLD)
    be = gf.Backend(device=0, options=opt)
    be.set_allreduce(gf.dist.torch_allreduce_hook(), rank, world)
    got = be.solve(snap, abi.MARGIN_OLD)
    again = be.solve(snap, abi.MARGIN_OLD)
    def pack(r, pre):
        o = {pre + "pose": r["state"]["pose"], pre + "sb": r["state"]["speed_bias"], pre + "feature": r["feature"],
             pre + "cost_history": np.array(r["summary"]["cost_history"]), pre + "accepted": np.array(r["summary"]["accepted"]),
             pre + "final_cost": r["summary"]["final_cost"], pre + "iterations": r["summary"]["iterations"],
             pre + "J0": r["prior"]["J0"], pre + "r0": r["prior"]["r0"], pre + "x0": r["prior"]["x0"]}
        return o
    res = {}
    res.update(pack(ref, "ref_")); res.update(pack(got, "got_")); res.update(pack(again, "again_"))
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
