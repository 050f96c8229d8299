"""Loader for the CPU oracle (oracle/, TEST INFRASTRUCTURE). Builds it with `make` on first use."""
import ctypes as C
import os
import subprocess

import numpy as np

from _gfbe_import import gf

abi = gf.abi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "libgfbe_oracle.so")


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return SO


def oracle_options():
    """The oracle's defaults (gfo_default_options): the product's, except that the square root of the new prior is the reference's
    eigen-decomposition (marg_sqrt = 0, marginalization_factor.cpp:294-302) unless a test or the like-for-like CPU baseline of
    bench.py asks for the pivoted LDL^T (marg_sqrt = 1)."""
    o = abi.default_options()
    o.marg_sqrt = 0
    return o


class Oracle(abi.CApi):
    prefix = "gfo_"

    def __init__(self, lib, opt=None):
        self.lib = abi.bind(lib, "gfo_")
        self.opt = opt or oracle_options()
        self.head = C.byref(self.opt)

    def with_options(self, **kw):
        o = oracle_options()
        for k, v in kw.items():
            setattr(o, k, v)
        return Oracle(self.lib, o)

    def linearize(self, snap):
        wh = snap if isinstance(snap, abi.WindowHolder) else abi.WindowHolder(snap)
        L = wh.n_feature
        H, g = np.zeros((abi.DENSE_DIM, abi.DENSE_DIM)), np.zeros(abi.DENSE_DIM)
        Hll, gl, Hpl = np.zeros(L), np.zeros(L), np.zeros((L, 73))
        cost = C.c_double(0)
        rc = self.lib.gfo_linearize(self.head, C.byref(wh.c), abi._pd(H), abi._pd(g), abi._pd(Hll), abi._pd(gl),
                                    abi._pd(Hpl), C.byref(cost))
        assert rc == 0
        return dict(H=H, g=g, Hll=Hll, gl=gl, Hpl=Hpl, cost=cost.value)

    def reanchor(self, before, after):
        b, a, o = abi.state_from_snapshot(before), abi.state_from_snapshot(after), abi.State()
        self.lib.gfo_reanchor(C.byref(b), C.byref(a), C.byref(o))
        return abi.state_to_dict(o)

    def marginalize(self, snap, flag):
        wh = snap if isinstance(snap, abi.WindowHolder) else abi.WindowHolder(snap)
        pr = abi.PriorHolder()
        A, b = np.zeros(abi.DENSE_DIM * abi.DENSE_DIM), np.zeros(abi.DENSE_DIM)
        rc = self.lib.gfo_marginalize(self.head, C.byref(wh.c), int(flag), C.byref(pr.c), abi._pd(A), abi._pd(b))
        if rc != 0 or not pr.c.valid:
            return None, None, None, rc
        n = pr.c.n
        return pr.to_dict(), A[: n * n].reshape(n, n).copy(), b[:n].copy(), rc

    def sqrt_info(self, cov):
        n = cov.shape[0]
        out = np.zeros((n, n))
        c = np.ascontiguousarray(cov, float)
        rc = self.lib.gfo_sqrt_info(abi._pd(c), abi._pd(out), n)
        assert rc == 0
        return out

    def sym_eig(self, A):
        n = A.shape[0]
        a = np.ascontiguousarray(A, float)
        w, V = np.zeros(n), np.zeros((n, n))
        self.lib.gfo_sym_eig(abi._pd(a), n, abi._pd(w), abi._pd(V))
        return w, V


_cached = None


def load():
    global _cached
    if _cached is None:
        build()
        _cached = Oracle(C.CDLL(SO))
    return _cached
