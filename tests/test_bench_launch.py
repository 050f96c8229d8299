"""bench.py's N-rank launch (`python bench.py --gpus N` must start N ranks itself) and the native RCCL hook's
contract, on CPU: (1) `bench.py --gpus 2 --launch-check` outside a launcher re-executes itself under torch.distributed.run,
both ranks rendezvous over gloo and rank 0 reports the aggregate (sum of units, max of elapsed); (2) libgfbe_rccl.so builds,
exports every symbol include/gfbe_rccl.h declares and refuses cleanly without a GPU."""
import ctypes as C
import json
import os
import re
import socket
import subprocess
import sys

from _gfbe_import import gf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_launches_its_own_ranks():
    env = dict(os.environ, GFBE_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--master-port", str(_free_port())],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout            # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["launch_check"] is True
    assert r["units"] == 100 + 101 and abs(r["elapsed"] - 1.5) < 1e-12      # SUM of units over ranks, MAX of elapsed


def test_bench_single_rank_launch_check_needs_no_launcher():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1


def test_rccl_hook_library_contract():
    so = gf.backend.build_rccl_hook()
    lib = C.CDLL(so)
    declared = set(re.findall(r"\b(gfbe_rccl_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", "gfbe_rccl.h")).read()))
    assert declared == set(gf.backend.RCCL_EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    # the hook has the signature of gfbe_allreduce_fn (include/gfbe.h): (user, device_ptr, n_doubles, hip_stream) -> status
    hdr = open(os.path.join(ROOT, "include", "gfbe_rccl.h")).read()
    assert "int32_t gfbe_rccl_allreduce(void *user, void *device_ptr, int64_t n_doubles, void *hip_stream);" in hdr
    assert "typedef int32_t (*gfbe_allreduce_fn)(void *user, void *device_ptr, int64_t n_doubles, void *hip_stream);" in open(os.path.join(ROOT, "include", "gfbe.h")).read()
    lib.gfbe_rccl_create.restype = C.c_int32
    lib.gfbe_rccl_last_error.restype = C.c_int32
    h, idb = C.c_void_p(), C.create_string_buffer(128)
    assert lib.gfbe_rccl_create(C.byref(h), idb, 0, 1, -1) == -1            # bad argument
    import torch
    if not torch.cuda.is_available():
        assert lib.gfbe_rccl_create(C.byref(h), idb, 0, 1, 0) == -2        # no GPU: refused, no crash, no handle
        assert not h.value
    assert lib.gfbe_rccl_last_error(None) == -1
    lib.gfbe_rccl_allreduce.restype = C.c_int32
    assert lib.gfbe_rccl_allreduce(None, None, 0, None) == -1               # null handle: refused with a status, no crash


def test_public_headers_are_strict_c99(tmp_path):
    """include/gfbe.h and include/gfbe_rccl.h together, as a C caller sees them."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "gfbe.h"\n#include "gfbe_rccl.h"\nint main(void) { return 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)], check=True)
