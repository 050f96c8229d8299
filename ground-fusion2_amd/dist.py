"""Multi-GPU helpers. One process per GPU under torch.distributed (backend "nccl" =
RCCL on ROCm; "gloo" in the CPU tests).

Primary mode — *replica / window sharding*: windows are independent units, rank r owns windows
r, r+world, ...; no data-path collective, only the timing barrier + max-over-ranks.

Secondary mode — *landmark sharding of one window* (BASELINE.json configs[2]): every rank uploads the
same window, evaluates the landmark tiles t with t % world == rank, eliminates them locally and
contributes partial normal equations; libgfbe calls the hook installed with gfbe_set_allreduce to sum
them in place: one all-reduce of [H | g | E | eg | cost] (~38.7k doubles per window) per linearisation
plus two 8-doubles-per-rank scalar exchanges per trust-region iteration. The hook returns a status: a failed
all-reduce makes the solve return GFBE_DEVICE_ERROR instead of un-reduced sums. The dense solve is then
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


class RcclHook:
    """libgfbe_rccl.so (include/gfbe_rccl.h): an ncclAllReduce on the solver's own stream, no interpreter on the path.
    The 128-byte unique id travels from rank 0 over the caller's torch.distributed group (one broadcast at start-up)."""

    def __init__(self, rank, world, device, group=None):
        import ctypes as C
        import os
        import torch
        import torch.distributed as dist
        from . import backend
        # One ROCm runtime per process: inside a Python process torch's bundled librccl.so.1 / libamdhip64 / libhsa-runtime64 are the
        # ones in use. Loaded here first (same SONAME as /opt/rocm's), it is the copy libgfbe_rccl.so's NEEDED entry resolves to —
        # /opt/rocm's RCCL would dlopen a second, uninitialised HSA runtime ("pfn_hsa_system_get_info failed", measured on the test box).
        tl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(tl):
            self._rccl = C.CDLL(tl, mode=C.RTLD_GLOBAL)
        self.lib = C.CDLL(backend._RCCL_SO)
        self.lib.gfbe_rccl_unique_id.restype = C.c_int32
        self.lib.gfbe_rccl_create.restype = C.c_int32
        self.lib.gfbe_rccl_last_error.restype = C.c_int32
        self.lib.gfbe_rccl_calls.restype = C.c_int64
        self.lib.gfbe_rccl_bytes.restype = C.c_int64
        self.lib.gfbe_rccl_comm_count.restype = C.c_int32
        idbuf = C.create_string_buffer(128)
        if rank == 0:
            rc = self.lib.gfbe_rccl_unique_id(idbuf)
            if rc != 0:
                raise RuntimeError("gfbe_rccl_unique_id failed: %d" % rc)
        if world > 1 or dist.is_initialized():       # (a one-rank communicator needs no rendezvous)
            t = torch.tensor(list(idbuf.raw), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0, group=group)
            idbuf = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
        self.h = C.c_void_p()
        rc = self.lib.gfbe_rccl_create(C.byref(self.h), idbuf, int(rank), int(world), int(device))
        if rc != 0:
            raise RuntimeError("gfbe_rccl_create failed: %d" % rc)
        self.fn_ptr = C.cast(self.lib.gfbe_rccl_allreduce, C.c_void_p).value

    def calls(self):
        return int(self.lib.gfbe_rccl_calls(self.h))

    def bytes(self):
        return int(self.lib.gfbe_rccl_bytes(self.h))

    def comm_count(self):
        return int(self.lib.gfbe_rccl_comm_count(self.h))

    def last_error(self):
        return int(self.lib.gfbe_rccl_last_error(self.h))

    def close(self):
        if self.h:
            self.lib.gfbe_rccl_destroy(self.h)
            self.h = None


def install_allreduce_hook(be, rank, world, prefer="native", group=None):
    """Landmark sharding on `be`: the native RCCL hook when asked for and usable (backend nccl, library built), else the
    torch.distributed callback. Returns the kind installed ("native-rccl" / "torch-<backend>")."""
    import torch.distributed as dist
    from . import backend
    import os
    solo = world == 1 and not dist.is_initialized()        # one rank, no process group: the native hook over a one-rank communicator
    if prefer == "native" and (solo or dist.get_backend(group) == "nccl") and os.path.exists(backend._RCCL_SO):
        hook = RcclHook(rank, world, be.device, group)
        be._rccl_hook = hook                       # keep the communicator alive as long as the back end
        be.set_allreduce_native(hook.fn_ptr, hook.h.value, rank, world)
        return "native-rccl"
    be.set_allreduce(torch_allreduce_hook(group), rank, world)
    return "torch-" + dist.get_backend(group)
