"""Loader of the gfx950 kernel library.  Fails loudly: there is NO fallback path."""
from __future__ import annotations

import ctypes
import os

from . import _cabi

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MST_HIP_LIB: developer override to load an A/B build of the same library (still no fallback)
LIB_PATH = os.environ.get("MST_HIP_LIB") or os.path.join(_PKG_ROOT, "lib", "libdiffmst_hip.so")
_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found - build it with `make -C diff-mst_amd/csrc` "
                "(hipcc --offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The mst package has no CPU fallback."
            )
        _lib = _cabi.bind(ctypes.CDLL(LIB_PATH))
    return _lib


def current_stream_ptr(device) -> ctypes.c_void_p:
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the raw handle without building a torch.cuda.Stream object (~10 us)
    if raw is not None:
        idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
        return ctypes.c_void_p(raw(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "mst (MI355X build): tensors must live on a ROCm device (got a CPU tensor); "
                "there is no CPU path in this package"
            )


def require_same_device(device, *tensors):
    for t in tensors:
        if t is not None and t.device != device:
            raise RuntimeError(f"mst (MI355X build): every tensor of a call must live on {device} (got one on {t.device})")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")
