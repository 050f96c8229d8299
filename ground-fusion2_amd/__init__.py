"""ground-fusion2_amd — MI355X-native sliding-window visual-inertial-wheel back end.

Host-side mirror of the ONE hot path of sjtuyinjie/Ground-Fusion2 this repo accelerates:
Estimator::optimization() (Ground-Fusion++/vins_estimator/src/estimator/estimator.cpp:2951-3698).
The product is csrc/libgfbe.so (hand-written HIP for gfx950 behind the C ABI of include/gfbe.h);
this package is the thin ctypes binding + synthetic-input generator. There is NO CPU fallback:
every compute entry point raises if the HIP library or a GPU is missing.
"""
from . import abi, synth, dist, stream  # noqa: F401
from . import backend  # noqa: F401
from .backend import Backend, BackendError, lib_path, build_native, WindowSet, DownloadBuffers, strip_visual  # noqa: F401

__version__ = "0.1.0"
