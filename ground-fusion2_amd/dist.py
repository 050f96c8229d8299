"""Multi-GPU helpers. One process per GPU under torch.distributed (backend "nccl" =
RCCL on ROCm; "gloo" in the CPU tests).

Primary mode — *replica / window sharding*: windows are independent units, rank r owns windows
r, r+world, ...; no data-path collective, only the timing barrier + max-over-ranks.

Secondary mode — *landmark sharding of one window* (BASELINE.json configs[2]): every rank uploads the
same window, evaluates the landmark tiles t with t % world == rank, eliminates them locally and
contributes partial normal equations; libgfbe calls the hook installed with gfbe_set_allreduce to sum
them in place: one all-reduce of [H | g | E | eg | cost] (~38.7k doubles per window) per linearisation
plus two 8-doubles-per-rank scalar exchanges per trust-region iteration. The dense solve is then
redundant (and bit-identical) on every rank. `torch_allreduce_hook` is that hook over torch.distributed.
"""
import numpy as np


def shard_indices(n_total, rank, world):
    """Round-robin ownership of n_total independent units."""
    return list(range(rank, n_total, world))


def aggregate_throughput(units_local, elapsed_local, dist=None):
    """(total units, max elapsed over ranks) — the bench.py contract: barrier, time, MAX over ranks."""
    if dist is None or not dist.is_initialized():
        return units_local, elapsed_local
    import torch
    t = torch.tensor([float(units_local), 0.0], dtype=torch.float64)
    m = torch.tensor([float(elapsed_local)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t, m = t.cuda(), m.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(round(float(t[0].item()))), float(m.item())


def shard_landmarks(snap, rank, world):
    """Window snapshot restricted to the landmarks rank owns (feature_index % world == rank).
    IMU / wheel / prior factors are kept on rank 0 only, so that summing the ranks' normal
    equations reproduces the unsharded ones."""
    s = dict(snap)
    keep = (np.asarray(snap["vis_feature_index"]) % world) == rank
    for k in list(snap):
        if k.startswith("vis_"):
            s[k] = np.asarray(snap[k])[keep]
    if rank != 0:
        s["imu"] = np.asarray(snap["imu"])[:0]
        s["imu_frame"] = np.asarray(snap["imu_frame"])[:0]
        if "wheel" in snap:
            s["wheel"] = np.asarray(snap["wheel"])[:0]
            s["wheel_frame"] = np.asarray(snap["wheel_frame"])[:0]
        s["prior"] = None
    return s


def reduced_system(lin, mu=0.0):
    """Schur-reduce one linearisation (H 182x182, g, per-landmark Hll/gl/Hpl[L,73]) onto the dense
    block; returns the packed [S | g | cost] vector that is all-reduced."""
    H, g = lin["H"].copy(), lin["g"].copy()
    ok = lin["Hll"] > 0
    w = np.where(ok, 1.0 / np.where(ok, lin["Hll"] * (1.0 + mu), 1.0), 0.0)
    hp = lin["Hpl"]
    H[:73, :73] -= (hp * w[:, None]).T @ hp
    g[:73] -= hp.T @ (w * lin["gl"])
    return np.concatenate([H.ravel(), g, [lin["cost"]]])


class _DevicePtr:
    """A device buffer of n float64 handed over by libgfbe, viewed through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def torch_allreduce_hook(group=None):
    """In-place sum all-reduce of a libgfbe device buffer over torch.distributed — the callable for
    Backend.set_allreduce(fn, rank, world). backend "nccl" (= RCCL over xGMI) reduces the device buffer
    directly on the solver's stream; "gloo" (tests: several ranks sharing one GPU) stages through the host."""
    import torch
    import torch.distributed as dist

    def hook(ptr, n, stream):
        t = torch.as_tensor(_DevicePtr(ptr, n), device="cuda")
        st = torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.default_stream()
        with torch.cuda.stream(st):
            if dist.get_backend(group) == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
    return hook
