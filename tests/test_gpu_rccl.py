"""The native RCCL hook (libgfbe_rccl.so, include/gfbe_rccl.h) on real hardware: a ONE-rank communicator on the test box's GPU.
gfbe_set_allreduce with world_size 1 still runs the landmark-sharded launch sequence — every partial slab goes through
ncclAllReduce on the solver's stream, which is then the identity — so the whole path (hook installation, the all-reduce of
[H | g | E | eg | cost], the scalar exchange blocks, the marginalisation partials, the final inverse-depth merge, the status
propagation) executes with RCCL itself in the loop and must reproduce the unsharded solve."""
import numpy as np
import pytest

from _gfbe_import import gf
from test_gpu_parity import window_with_prior

abi, synth = gf.abi, gf.synth
pytestmark = pytest.mark.gpu


def test_one_rank_rccl_all_reduce_on_the_solver_stream(oracle):
    _, snap = window_with_prior(oracle, 91, 600)
    plain = gf.Backend(device=0)
    ref = plain.solve(snap, abi.MARGIN_OLD)
    be = gf.Backend(device=0)
    kind = gf.dist.install_allreduce_hook(be, 0, 1, prefer="native")
    assert kind == "native-rccl"
    hook = be._rccl_hook
    got = be.solve(snap, abi.MARGIN_OLD)
    assert hook.last_error() == 0
    n_calls = hook.calls()
    # per linearisation one system slab + one mu-retry slab + one scalar exchange, per iteration one more exchange, the
    # marginalisation's two partial slabs and the inverse-depth merge: dozens of collectives per solve
    assert n_calls >= 3 * got["summary"]["iterations"] + 3, n_calls
    # identity all-reduces: the sharded launch sequence on one rank is the unsharded solve up to the summation order of the
    # exchange blocks
    assert got["summary"]["accepted"] == ref["summary"]["accepted"] and got["summary"]["iterations"] == ref["summary"]["iterations"]
    np.testing.assert_allclose(got["summary"]["cost_history"], ref["summary"]["cost_history"], rtol=1e-9)
    assert np.abs(got["state"]["pose"] - ref["state"]["pose"]).max() < 1e-9
    np.testing.assert_allclose(got["feature"], ref["feature"], rtol=1e-8, atol=1e-12)
    Ag, Ar = got["prior"]["J0"].T @ got["prior"]["J0"], ref["prior"]["J0"].T @ ref["prior"]["J0"]
    assert np.abs(Ag - Ar).max() < 1e-8 * np.abs(Ar).max()
    # a batch of windows through the same communicator, twice (the communicator is reused)
    res = be.solve_batch([snap, snap], abi.MARGIN_NONE)
    assert res[0]["summary"] == res[1]["summary"]
    assert hook.calls() > n_calls and hook.last_error() == 0
    assert hook.comm_count() == 1 and hook.bytes() >= 8 * (hook.calls() - n_calls)        # (ncclCommCount of the communicator; what bench.py --shard-landmarks reports)
    be.close()                       # destroys the context, then the communicator
    assert be._rccl_hook is None
    plain.close()


def test_failing_hook_makes_the_solve_fail(oracle):
    """The hook's status is propagated: un-reduced partial sums are never returned as a result."""
    scn = synth.Scenario(seed=92, n_landmarks=200, use_wheel=True)
    be = gf.Backend(device=0)

    def broken(ptr, n, stream):
        raise RuntimeError("no collective today")
    be.set_allreduce(broken, 0, 1)
    with pytest.raises(gf.backend.BackendError, match="all-reduce hook failed"):
        be.solve(scn.window(0), abi.MARGIN_NONE)
    be.close()
