"""ctypes binding of the product library csrc/libgfbe.so (HIP, gfx950) — the host-side mirror of
Estimator::optimization() (estimator.cpp:2951-3698) over the C ABI of include/gfbe.h.

There is deliberately no CPU path here: if the library is missing, a symbol is missing, or no GPU
is visible, the call raises BackendError (gfbe_status GFBE_NO_DEVICE). The CPU oracle lives in
oracle/ and is only ever loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# GFBE_LIB: an alternative build of the library (tools/diag_variants.py compares kernel variants built side by side)
_SO = os.environ.get("GFBE_LIB") or os.path.join(_CSRC, "libgfbe.so")

# Every symbol include/gfbe.h declares (checked by tests/test_abi.py on CPU).
EXPORTS = [
    "gfbe_default_options", "gfbe_options_size", "gfbe_create", "gfbe_destroy", "gfbe_last_error", "gfbe_create_note", "gfbe_host_times", "gfbe_version", "gfbe_set_stream",
    "gfbe_feature_count", "gfbe_visual_factor_count", "gfbe_build_visual_factors", "gfbe_set_depth",
    "gfbe_eval_factors", "gfbe_preintegrate_imu", "gfbe_preintegrate_wheel",
    "gfbe_solve_window", "gfbe_solve_batch",
    "gfbe_batch_upload", "gfbe_batch_solve", "gfbe_batch_download", "gfbe_batch_free",
    "gfbe_profile_enable", "gfbe_profile_count", "gfbe_profile_get", "gfbe_profile_reset",
    "gfbe_set_allreduce", "gfbe_debug_timing", "gfbe_debug_vector",
    "gfbe_ftab_default_options", "gfbe_ftab_create", "gfbe_ftab_destroy", "gfbe_ftab_add_frame",
    "gfbe_ftab_remove_back_shift_depth", "gfbe_ftab_remove_back", "gfbe_ftab_remove_front", "gfbe_ftab_remove_outlier",
    "gfbe_ftab_remove_failures", "gfbe_ftab_clear_depth", "gfbe_ftab_set_depth", "gfbe_ftab_get_depth_vector",
    "gfbe_ftab_triangulate", "gfbe_ftab_check_outliers", "gfbe_ftab_size", "gfbe_ftab_download", "gfbe_slide_window_state",
    "gfbe_pg_eval", "gfbe_pg_solve", "gfbe_lio_linearize", "gfbe_batch_upload_tables", "gfbe_batch_feature_count",
    "gfbe_plane_eval", "gfbe_anchor_eval", "gfbe_orientation_subset_plus", "gfbe_gnss_eval",
]


class BackendError(RuntimeError):
    pass


def lib_path():
    return _SO


def sources():
    return sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".cpp")))


DIAG_SO = os.path.join(_CSRC, "libgfbe_diag.so")


def build_native(force=False, verbose=False, out=None, extra_flags=None, fp_contract="off", diag=False):
    """hipcc --offload-arch=gfx950 -> csrc/libgfbe.so (cross-compiles without a GPU). Every source is compiled to its own
    object (in parallel, only when it is older than the source or a header) and the objects are linked: a one-file change
    rebuilds in seconds. `out` / `extra_flags` / `fp_contract`: side-by-side variant builds (tools/diag_variants.py)."""
    from concurrent.futures import ThreadPoolExecutor
    if diag:
        out = out or DIAG_SO
        extra_flags = list(extra_flags or []) + ["-DGFBE_DIAG=1"]
    so = out or os.path.join(_CSRC, "libgfbe.so")
    srcs = sources()
    hdrs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".hpp"))]
    hdrs.append(os.path.join(os.path.dirname(_HERE), "include", "gfbe.h"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = os.environ.get("GFBE_EXTRA_FLAGS", "").split() if extra_flags is None else list(extra_flags)
    cflags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=" + fp_contract, "-munsafe-fp-atomics",
              "-Wall", "-Wno-unused-function", "-pthread", "-I", os.path.join(os.path.dirname(_HERE), "include")] + flags
    tag = "default" if (out is None and not flags and fp_contract == "off") else os.path.splitext(os.path.basename(so))[0]
    odir = os.path.join(_CSRC, "build", tag)
    os.makedirs(odir, exist_ok=True)
    stamp = os.path.join(odir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(cflags):
        force = True
    hnew = max(os.path.getmtime(h) for h in hdrs)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(odir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hnew):
            jobs.append([hipcc] + cflags + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(o) for o in objs):
        return so

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", so] + objs)
    with open(stamp, "w") as f:
        f.write(" ".join(cflags))
    return so


_RCCL_SO = os.path.join(_CSRC, "libgfbe_rccl.so")
RCCL_EXPORTS = ["gfbe_rccl_unique_id", "gfbe_rccl_create", "gfbe_rccl_destroy", "gfbe_rccl_allreduce", "gfbe_rccl_last_error", "gfbe_rccl_calls",
                "gfbe_rccl_bytes", "gfbe_rccl_comm_count"]


def build_rccl_hook(force=False, verbose=False):
    """csrc/rccl/gfbe_rccl_hook.cpp -> csrc/libgfbe_rccl.so (the native all-reduce hook of include/gfbe_rccl.h; links librccl)."""
    src = os.path.join(_CSRC, "rccl", "gfbe_rccl_hook.cpp")
    deps = [src, os.path.join(os.path.dirname(_HERE), "include", "gfbe_rccl.h")]
    if not force and os.path.exists(_RCCL_SO) and all(os.path.getmtime(_RCCL_SO) >= os.path.getmtime(d) for d in deps):
        return _RCCL_SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", "/opt/rocm/include",
           "-o", _RCCL_SO, src, "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return _RCCL_SO


class Backend(abi.CApi):
    prefix = "gfbe_"

    def __init__(self, device=0, options=None, so=None):
        _SO = so or globals()["_SO"]          # (so: another build of the library, e.g. DIAG_SO)
        if not os.path.exists(_SO):
            raise BackendError("HIP extension %s is missing: run __graft_entry__.build() (no CPU fallback)" % _SO)
        # the solver drives up to eight streams: the HIP runtime's hardware-queue count is the CALLER's to set, before HIP
        # initialises in this process; a value the embedding application already chose is left alone
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        self.lib = abi.bind(C.CDLL(_SO), "gfbe_")
        self.opt = options or abi.default_options()
        self.ctx = C.c_void_p()
        self.lib.gfbe_create.restype = abi.c_i
        self.lib.gfbe_last_error.restype = C.c_char_p
        self.lib.gfbe_last_error.argtypes = [C.c_void_p]
        self.lib.gfbe_version.restype = C.c_char_p
        rc = self.lib.gfbe_create(C.byref(self.ctx), int(device), C.byref(self.opt))
        if rc != abi.OK:
            raise BackendError("gfbe_create(device=%d) failed with status %d (%s)" % (device, rc, self._err()))
        self.lib.gfbe_create_note.restype = C.c_char_p
        self.lib.gfbe_create_note.argtypes = [C.c_void_p]
        self.create_note = (self.lib.gfbe_create_note(self.ctx) or b"").decode()   # (a remark, not an error: e.g. fewer than eight hardware queues configured)
        self.head = self.ctx
        self.device = device
        for name in ("batch_upload", "batch_solve", "batch_download", "solve_batch", "profile_enable",
                     "profile_get", "set_stream", "set_allreduce", "eval_factors", "solve_window"):
            getattr(self.lib, "gfbe_" + name).restype = abi.c_i
        self.lib.gfbe_batch_free.restype = None
        self.lib.gfbe_profile_count.restype = abi.c_i
        self._cb = None

    def _err(self):
        try:
            return (self.lib.gfbe_last_error(self.ctx) or b"").decode()
        except Exception:
            return "?"

    def check(self, rc, what):
        if rc not in (abi.OK, abi.NO_CONVERGENCE):
            raise BackendError("gfbe_%s failed with status %d: %s" % (what, rc, self._err()))
        return rc

    def close(self):
        if self.ctx:
            self.lib.gfbe_destroy(self.ctx)
            self.ctx = C.c_void_p()
        hook = getattr(self, "_rccl_hook", None)      # (the native all-reduce hook's communicator lives as long as the context)
        if hook is not None:
            hook.close()
            self._rccl_hook = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def version(self):
        return self.lib.gfbe_version().decode()

    def set_stream(self, stream_ptr):
        self.check(self.lib.gfbe_set_stream(self.ctx, C.c_void_p(stream_ptr)), "set_stream")

    # ---- device-resident batch
    def batch_upload(self, snaps):
        if isinstance(snaps, WindowSet):      # prebuilt pointer array: no per-call Python work (the end-to-end loops of bench.py)
            holders, arr = snaps.holders, snaps.arr
        else:
            holders = [s if isinstance(s, abi.WindowHolder) else abi.WindowHolder(s) for s in snaps]
            arr = (C.POINTER(abi.Window) * len(holders))(*[C.pointer(h.c) for h in holders])
        batch = C.c_void_p()
        self.check(self.lib.gfbe_batch_upload(self.ctx, len(holders), arr, C.byref(batch)), "batch_upload")
        return Batch(self, batch, holders)

    def batch_upload_tables(self, tables, snaps):
        """snaps: window snapshots WITHOUT visual factors (state, pre-integrations, prior, flags); the landmarks of window w
        come from table w of `tables` (abi.FeatureTables on this library) without leaving the device."""
        if isinstance(snaps, WindowSet):
            batch = C.c_void_p()
            self.lib.gfbe_batch_upload_tables.restype = abi.c_i
            self.check(self.lib.gfbe_batch_upload_tables(self.ctx, tables.h, len(snaps.holders), snaps.arr, C.byref(batch)), "batch_upload_tables")
            return Batch(self, batch, snaps.holders)
        holders = []
        for s in snaps:
            s = dict(s)
            for k in ("vis_feature_index", "vis_imu_i", "vis_imu_j", "vis_pts_i", "vis_pts_j", "vis_vel_i", "vis_vel_j", "vis_td_i", "vis_td_j"):
                s[k] = np.zeros(0)
            s["para_feature"], s["feature_const"] = np.zeros(0), np.zeros(0, np.uint8)
            holders.append(abi.WindowHolder(s))
        arr = (C.POINTER(abi.Window) * len(holders))(*[C.pointer(h.c) for h in holders])
        batch = C.c_void_p()
        self.lib.gfbe_batch_upload_tables.restype = abi.c_i
        self.check(self.lib.gfbe_batch_upload_tables(self.ctx, tables.h, len(holders), arr, C.byref(batch)), "batch_upload_tables")
        for w, h in enumerate(holders):
            h.n_feature_override = int(self.lib.gfbe_batch_feature_count(batch, w))
        return Batch(self, batch, holders)

    def solve_batch(self, snaps, margin_flag=abi.MARGIN_NONE):
        b = self.batch_upload(snaps)
        try:
            b.solve(margin_flag)
            return b.download()
        finally:
            b.free()

    # ---- profiling hooks
    def profile_enable(self, on=True):
        self.check(self.lib.gfbe_profile_enable(self.ctx, int(on)), "profile_enable")

    def profile_reset(self):
        self.lib.gfbe_profile_reset(self.ctx)

    def profile(self):
        out = []
        for i in range(self.lib.gfbe_profile_count(self.ctx)):
            name = C.c_char_p()
            launches = C.c_int64()
            ms = C.c_double()
            by = C.c_double()
            self.lib.gfbe_profile_get(self.ctx, i, C.byref(name), C.byref(launches), C.byref(ms), C.byref(by))
            out.append(dict(name=name.value.decode(), launches=launches.value, total_ms=ms.value, bytes=by.value))
        return out

    def host_times(self):
        """gfbe_host_times: ms of the calling thread in the last upload (packing | the rest) and the last download (waiting for the device | unpacking)."""
        out = (C.c_double * 4)()
        self.lib.gfbe_host_times.restype = None
        self.lib.gfbe_host_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self.lib.gfbe_host_times(self.ctx, out)
        return dict(upload_pack_ms=out[0], upload_rest_ms=out[1], download_wait_ms=out[2], download_unpack_ms=out[3])

    # ---- multi-GPU landmark sharding: the native hook (libgfbe_rccl.so: ncclAllReduce on the solver's stream)
    def set_allreduce_native(self, fn_ptr, user_ptr, rank, world_size):
        """fn_ptr: address of a gfbe_allreduce_fn (e.g. gfbe_rccl_allreduce), user_ptr: its handle."""
        CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
        self._cb = C.cast(fn_ptr, CB)
        self.check(self.lib.gfbe_set_allreduce(self.ctx, self._cb, C.c_void_p(user_ptr), int(rank), int(world_size)), "set_allreduce")

    # ---- the same through a Python callable (torch.distributed in the caller; the tests' gloo flavour)
    def set_allreduce(self, fn, rank, world_size):
        CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

        def call(user, ptr, n, stream):      # (0, or 1 when the Python hook raised: the solve then returns GFBE_DEVICE_ERROR)
            try:
                fn(ptr, n, stream)
                return 0
            except Exception:                # noqa: BLE001 — an exception must not unwind through the C library
                import traceback
                traceback.print_exc()
                return 1
        self._cb = CB(call) if fn is not None else C.cast(None, CB)
        self.check(self.lib.gfbe_set_allreduce(self.ctx, self._cb, None, int(rank), int(world_size)), "set_allreduce")


def strip_visual(snap):
    """Window snapshot without its visual part (what gfbe_batch_upload_tables reads from the host)."""
    s = dict(snap)
    for k in ("vis_feature_index", "vis_imu_i", "vis_imu_j", "vis_pts_i", "vis_pts_j", "vis_vel_i", "vis_vel_j", "vis_td_i", "vis_td_j"):
        s[k] = np.zeros(0)
    s["para_feature"], s["feature_const"] = np.zeros(0), np.zeros(0, np.uint8)
    return s


class WindowSet:
    """A list of windows with the ctypes pointer array gfbe_batch_upload takes, built once (bench.py's end-to-end loops
    re-upload the same host structures every step; the library re-packs and re-copies them every time)."""

    def __init__(self, snaps):
        self.holders = [s if isinstance(s, abi.WindowHolder) else abi.WindowHolder(s) for s in snaps]
        self.arr = (C.POINTER(abi.Window) * len(self.holders))(*[C.pointer(h.c) for h in self.holders])


class DownloadBuffers:
    """Pre-allocated outputs of gfbe_batch_download for n windows of up to max_features landmarks."""

    def __init__(self, n, max_features):
        self.n = n
        self.states = (abi.State * n)()
        self.feats = np.zeros((n, max(int(max_features), 1)))
        self.fptr = (abi.PD * n)(*[abi._pd(self.feats[k]) for k in range(n)])
        self.priors = [abi.PriorHolder() for _ in range(n)]
        self.pptr = (C.POINTER(abi.Prior) * n)(*[C.pointer(p.c) for p in self.priors])
        self.sums = (abi.Summary * n)()


class Batch:
    def __init__(self, be, handle, holders):
        self.be, self.h, self.holders = be, handle, holders
        self.n = len(holders)

    def solve(self, margin_flag=abi.MARGIN_NONE):
        self.be.check(self.be.lib.gfbe_batch_solve(self.be.ctx, self.h, int(margin_flag)), "batch_solve")

    def download(self, raise_on_failure=True):
        """Per-window results. raise_on_failure=False: a window whose linear solves all failed (status NUMERICAL_FAILURE) does not
        raise — every window's own outcome is in its "status" (the batch's windows are independent)."""
        n = self.n
        states = (abi.State * n)()
        feats = [np.zeros(getattr(h, "n_feature_override", h.n_feature)) for h in self.holders]
        fptr = (abi.PD * n)(*[abi._pd(f) for f in feats])
        priors = [abi.PriorHolder() for _ in range(n)]
        pptr = (C.POINTER(abi.Prior) * n)(*[C.pointer(p.c) for p in priors])
        sums = (abi.Summary * n)()
        rc = self.be.lib.gfbe_batch_download(self.be.ctx, self.h, states, fptr, pptr, sums)
        if raise_on_failure or rc != abi.NUMERICAL_FAILURE:
            self.be.check(rc, "batch_download")
        out = []
        for k in range(n):
            out.append(dict(state=abi.state_to_dict(states[k]), feature=feats[k],
                            prior=priors[k].to_dict() if priors[k].c.valid else None,
                            summary=abi.summary_to_dict(sums[k]), perf=abi.summary_perf(sums[k]), status=sums[k].status))
        return out

    def download_into(self, bufs):
        """gfbe_batch_download into pre-allocated buffers (no per-window Python work); returns the status."""
        return self.be.check(self.be.lib.gfbe_batch_download(self.be.ctx, self.h, bufs.states, bufs.fptr, bufs.pptr, bufs.sums), "batch_download")

    def debug_timing(self, w=0):
        out = np.zeros(32)
        self.be.lib.gfbe_debug_timing(self.be.ctx, self.h, int(w), abi._pd(out))
        return out

    def debug_vector(self, which, w=0):
        out = np.zeros(abi.DENSE_DIM)
        self.be.check(self.be.lib.gfbe_debug_vector(self.be.ctx, self.h, int(w), int(which), abi._pd(out)), "debug_vector")
        return out

    def free(self):
        if self.h:
            self.be.lib.gfbe_batch_free(self.be.ctx, self.h)
            self.h = C.c_void_p()
