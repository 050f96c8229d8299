"""Contexts of one process: gfbe_destroy hands the context's HIP streams back to a process-wide pool and the next gfbe_create takes them
(gfbe_host.cpp: stream_acquire / stream_release — a context created after another one's streams had been destroyed used to run 7 % slower).
What that must not change: two contexts alive at once have streams of their own, a context created after another one was destroyed works
like the first one did, and every one of them returns the same bits — for one window and for a throughput batch whose parts hold streams
of their own."""
import numpy as np
import pytest

from _gfbe_import import gf

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


def _digest(results):
    return [(r["summary"]["iterations"], tuple(r["summary"]["cost_history"]), r["state"]["pose"].tobytes(), r["feature"].tobytes(),
             None if r["prior"] is None else r["prior"]["J0"].tobytes()) for r in results]


def test_contexts_side_by_side_and_one_after_the_other():
    scn = synth.Scenario(seed=41, n_landmarks=260, use_wheel=True)
    first = scn.window(0)
    be1 = gf.Backend(device=0)
    r0 = be1.solve(first, abi.MARGIN_OLD)
    snap = scn.window(1, state=synth.shift_state_for_next_window(scn, r0["state"], 1), prior=r0["prior"])
    batch = [snap if i % 3 else first for i in range(70)]          # (a throughput batch: two kernel sets, both with and without a prior)
    ref1, refb = _digest([be1.solve(snap, abi.MARGIN_OLD)]), _digest(be1.solve_batch(batch, abi.MARGIN_OLD))
    be2 = gf.Backend(device=0)                                      # alive beside be1: streams of its own
    up1, up2 = be1.batch_upload(batch), be2.batch_upload(batch)
    up1.solve(abi.MARGIN_OLD); up2.solve(abi.MARGIN_OLD)            # both enqueued before either is waited for
    assert _digest(up2.download()) == refb and _digest(up1.download()) == refb
    up1.free(); up2.free()
    assert _digest([be2.solve(snap, abi.MARGIN_OLD)]) == ref1
    be1.close()
    be3 = gf.Backend(device=0)                                      # takes be1's streams from the pool while be2 still holds its own
    assert _digest([be3.solve(snap, abi.MARGIN_OLD)]) == ref1
    assert _digest(be3.solve_batch(batch, abi.MARGIN_OLD)) == refb
    assert _digest(be2.solve_batch(batch, abi.MARGIN_OLD)) == refb
    be2.close(); be3.close()
    for _ in range(3):                                              # create / destroy in a row: the same streams every time
        be = gf.Backend(device=0)
        assert _digest([be.solve(snap, abi.MARGIN_OLD)]) == ref1
        be.close()


def test_a_split_batch_returns_its_lanes():
    """A batch of >= 2048 windows runs as four parts, three of them on stream pairs of their own (make_lane): freed with the batch into the
    context's lane pool, handed back to the process's pool with the context, taken again by the next context's split batch."""
    scn = synth.Scenario(seed=42, n_landmarks=64, use_wheel=False)
    wins = [scn.window(0)] * 2048
    ref = None
    for _ in range(2):
        be = gf.Backend(device=0)
        b = be.batch_upload(wins)
        b.solve(abi.MARGIN_OLD)
        res = b.download()
        b.free()
        be.close()
        dg = _digest(res[:3] + res[-3:])
        assert len(set(dg)) == 1                                    # (the same window in every place of every part)
        ref = ref or dg
        assert dg == ref
