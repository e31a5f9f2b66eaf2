"""Loads the HIP engine (csrc/libazgym_hip.so).  There is NO CPU fallback: if the library is missing or a GPU
is not present, constructing an engine raises."""
import ctypes as C
import os

from . import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
# AZG_HIP_LIB: diagnostic builds only (e.g. the -DAZG_STAMPS library used by tools/phase_profile.py)
LIB_PATH = os.environ.get("AZG_HIP_LIB") or os.path.join(HERE, "csrc", "libazgym_hip.so")

_lib = None
_fns = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    global _lib, _fns
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C alphazero_gym_amd/csrc` (hipcc, gfx950). This package has no CPU fallback."
            )
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's), and the
        # dynamic linker gives every later library whichever copy was loaded first.  If this engine came first and pulled in the
        # system runtime, a later torch.cuda initialisation fails ("No HIP GPUs are available"); so torch, when present, goes first.
        # (Round 3 tried loading torch's runtime copy by path instead of importing torch: engine-first then worked in one GPU run
        # and hung in torch's CUDA initialisation in the next -- reverted.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _fns = _capi.bind(_lib, "azg_")
        ver = _fns["abi_version"]()
        if ver != _capi.ABI_VERSION:
            raise NativeLibraryMissing(f"ABI version mismatch: library {ver}, binding {_capi.ABI_VERSION}")
        _lib.azg_math_selftest.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t]
    return _lib


def fns():
    lib()
    return _fns


class HipEngine(_capi.Engine):
    """Batched MCTS engine on one MI355X."""

    def __init__(self, **kw):
        super().__init__(fns(), **kw)


def math_selftest(fn_id, x, device_id=0):
    import numpy as np

    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    rc = lib().azg_math_selftest(device_id, fn_id, _capi._ptr(x, C.c_double), _capi._ptr(out, C.c_double), x.size)
    if rc != 0:
        raise _capi.EngineError(rc, "azg_math_selftest failed")
    return out
