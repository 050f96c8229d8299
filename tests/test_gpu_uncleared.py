"""Nothing reads what nobody wrote. A batch's slab is cleared only where the kernels expect zeros (gfbe_batch_upload: 1 of its ~8 MB
per 2k-landmark window); everything else — the assembled system, the landmark rows past a track's length, the marginalisation's
work matrices, the download staging ... — must be written before it is read, and a value that is loaded speculatively must be
replaced, not multiplied by zero. The test hook GFBE_POISON_UNCLEARED=1 (diagnostics build of the library only) fills the not-cleared part with NaN bit patterns at every
upload: the results must not change by a bit. (It found k_solve_chain's wide rows multiplying unwritten entries of H by a zero mask:
harmless while the slab's previous contents are finite, a spurious "linear solve failed" when they are not.)"""
import os

import numpy as np
import pytest

from _gfbe_import import gf
import gnss_window_cases as gw

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


def _digest(results):
    out = []
    for r in results:
        s = r["summary"]
        out.append((s["iterations"], s["termination"], tuple(s["cost_history"]), r["state"]["pose"].tobytes(), r["feature"].tobytes(),
                    None if r["prior"] is None else (r["prior"]["J0"].tobytes(), r["prior"]["r0"].tobytes(), tuple(r["prior"]["block_id"].tolist()))))
    return out


def test_results_do_not_depend_on_the_not_cleared_part_of_the_slab():
    # the hook exists in the diagnostics build only (libgfbe_diag.so, -DGFBE_DIAG=1: the same sources, the environment hooks compiled in)
    assert os.path.exists(gf.backend.DIAG_SO), "run __graft_entry__.build()"
    be = gf.Backend(device=0, so=gf.backend.DIAG_SO)
    scn = synth.Scenario(seed=11, n_landmarks=300, use_wheel=True)
    r0 = be.solve(scn.window(0), abi.MARGIN_OLD)
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    cases = {"one window with prior": [snap],
             "throughput batch (40 windows, with and without prior)": [snap if i % 2 else scn.window(0) for i in range(40)],
             "GNSS window": [gw.gnss_window(seed=81, L=150, n_per_frame=8)[2]],
             "window without landmarks": [synth.Scenario(seed=12, n_landmarks=0, use_wheel=True).window(0)],
             "ragged tracks, constant landmarks": [dict(synth.Scenario(seed=13, n_landmarks=700, use_wheel=False).window(0),
                                                        feature_const=(np.arange(700) % 3 == 0).astype(np.uint8))]}
    old = os.environ.pop("GFBE_POISON_UNCLEARED", None)
    try:
        for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
            want = {k: _digest(be.solve_batch(v, flag)) for k, v in cases.items()}
            os.environ["GFBE_POISON_UNCLEARED"] = "1"
            for k, v in cases.items():
                assert _digest(be.solve_batch(v, flag)) == want[k], (k, flag)
            os.environ.pop("GFBE_POISON_UNCLEARED")
    finally:
        os.environ.pop("GFBE_POISON_UNCLEARED", None)
        if old is not None:
            os.environ["GFBE_POISON_UNCLEARED"] = old


def test_merged_launches_equal_the_one_kernel_per_launch_sequence():
    """A small batch runs an iteration in six launches (k_schur + k_visblock_small in one launch, k_step / k_candidate
    at the tail of k_lm_step, k_accept at the tail of k_lin_small<1>, the marginalisation's four linearisation launches in two);
    with the library's per-kernel profiling on, every kernel is a launch of its own again. Same code, same order of every sum:
    the same bits — with and without landmarks, GNSS, LiDAR-free; for one window and for a batch of seven."""
    be = gf.Backend(device=0)
    scn = synth.Scenario(seed=21, n_landmarks=500, use_wheel=True)
    r0 = be.solve(scn.window(0), abi.MARGIN_OLD)
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    cases = {"one window": [snap], "seven windows": [snap if i % 2 else scn.window(0) for i in range(7)],
             "GNSS window": [gw.gnss_window(seed=85, L=120, n_per_frame=4, anchor=True)[2]],
             "no landmarks": [synth.Scenario(seed=22, n_landmarks=0, use_wheel=True).window(0)]}
    for flag in (abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW):
        merged = {k: _digest(be.solve_batch(v, flag)) for k, v in cases.items()}
        be.profile_enable(True)
        try:
            for k, v in cases.items():
                assert _digest(be.solve_batch(v, flag)) == merged[k], (k, flag)
        finally:
            be.profile_enable(False)
            be.profile_reset()


def test_compact_assembly_table_equals_the_full_one_entry_by_entry():
    """The compact assembly table (csrc/gfbe_kernels.hip: k_asm_compact) lists the entries of H's lower triangle a factor of a
    reference-structured window can reach, by a hand-written structural predicate; the others are never written and read as the zeros
    the upload left there. A factor or prior block coupling dims outside the predicate would lose entries SILENTLY. The diagnostics
    switch GFBE_ASM_FULL=1 assembles the same batch through the full table (every entry of the 187 x 187 triangle written): H must
    agree entry by entry — wheel, plane, anchor, LiDAR and prior windows, throughput batch included — and so must the results."""
    import plane_cases as pc
    assert os.path.exists(gf.backend.DIAG_SO), "run __graft_entry__.build()"
    be = gf.Backend(device=0, so=gf.backend.DIAG_SO)
    scn = synth.Scenario(seed=21, n_landmarks=300, use_wheel=True)
    r0 = be.solve(scn.window(0), abi.MARGIN_OLD)
    with_prior = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    pscn, psnap = pc.plane_window(seed=73, L=150, anchor=True)
    pres = be.solve(psnap, abi.MARGIN_OLD)
    plane_next = pc.next_plane_window(pscn, psnap, pres)            # prior with the plane blocks + plane factors + anchor-free
    lio = dict(scn.window(0), lio=synth.lidar_block(scn, 0, n=300, seed=5, frame=7))
    no_wheel = synth.Scenario(seed=22, n_landmarks=200, use_wheel=False).window(0)
    cases = {"wheel + prior": [with_prior], "plane + anchor": [psnap], "plane + prior with plane blocks": [plane_next], "lidar": [lio],
             "no wheel": [no_wheel], "throughput batch": [with_prior if i % 3 == 0 else (plane_next if i % 3 == 1 else no_wheel) for i in range(36)]}
    nc = 187
    old = os.environ.pop("GFBE_ASM_FULL", None)

    def run(snaps):
        b = be.batch_upload(snaps)
        try:
            b.solve(abi.MARGIN_OLD)
            res = b.download()
            H = [np.array([b.debug_vector(1000 + a, w)[:nc] for a in range(nc)]) for w in range(min(len(snaps), 3))]
            return _digest(res), [np.tril(h) for h in H]
        finally:
            b.free()
    try:
        for name, snaps in cases.items():
            dig_c, H_c = run(snaps)
            os.environ["GFBE_ASM_FULL"] = "1"
            dig_f, H_f = run(snaps)
            os.environ.pop("GFBE_ASM_FULL")
            for hc, hf in zip(H_c, H_f):
                assert np.count_nonzero(hf) > 1000, name
                np.testing.assert_array_equal(hc, hf, err_msg=name)          # (the last linearisation's H: same bits, zeros where the compact table has no entry)
            assert dig_c == dig_f, name
            # (end of round 6) the branch-free entry loops of the assembly — asm_H_tp, the early table staging, gather_g_dense_tp: GFBE_ASM_TP —
            # against their form of rounds 4-6 (GFBE_ASM_LEGACY=1, read at upload by the diagnostics build): the same terms in the same
            # order, so H entry by entry and every output bit for bit
            os.environ["GFBE_ASM_LEGACY"] = "1"
            dig_l, H_l = run(snaps)
            os.environ.pop("GFBE_ASM_LEGACY")
            for hc, hl in zip(H_c, H_l):
                np.testing.assert_array_equal(hc, hl, err_msg=name + " (legacy assembly)")
            assert dig_c == dig_l, name + " (legacy assembly)"
    finally:
        os.environ.pop("GFBE_ASM_FULL", None)
        os.environ.pop("GFBE_ASM_LEGACY", None)
        if old is not None:
            os.environ["GFBE_ASM_FULL"] = old
