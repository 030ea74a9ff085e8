"""Loader of ``libptgpu.so``, the HIP implementation behind the C ABI of ``include/ptgpu.h``.

There is no CPU fallback: if the shared library is missing or fails to load
this raises, and so does every entry point when no gfx950 device is usable.
"""
from __future__ import annotations

import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTGPU_LIB") or os.path.join(_HERE, "lib", "libptgpu.so")  # PTGPU_LIB: alternative build (kernel experiments)
_lib = None


class PtgpuError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"ptgpu error {code}: {message}")
        self.code = code


def load() -> C.CDLL:
    """Load libptgpu.so (once).  torch is imported first when available so that the process has a
    single HIP runtime: torch ships its own libamdhip64 with the same SONAME."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    abi.declare_ptgpu(lib)
    _lib = lib
    return lib
