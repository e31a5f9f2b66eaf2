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


class HipRuntimeConflict(NativeLibraryMissing):
    """Two different libamdhip64 files would end up (or are) mapped in this process."""


def mapped_hip_runtimes():
    """Real paths of the libamdhip64 files mapped into this process (Linux: /proc/self/maps)."""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line and "/" in line:
                    found.add(os.path.realpath(line[line.index("/"):].strip()))
    except OSError:
        pass
    return found


def _check_load_order():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's) and load it by path.  If some other HIP
    runtime is mapped already -- an embedding application loaded this engine, or another HIP library, before anything imported
    torch -- importing torch now gives the process two runtimes, and torch.cuda then reports no GPU or hangs.  Say so instead."""
    import importlib.util
    import sys
    have = mapped_hip_runtimes()
    if not have or "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled) and os.path.realpath(bundled) not in have and os.environ.get("AZG_ALLOW_MULTI_HIP") != "1":
        raise HipRuntimeConflict(
            f"a HIP runtime is already mapped in this process ({', '.join(sorted(have))}) and PyTorch, which has not been imported yet, "
            f"bundles another one ({bundled}): with both loaded torch.cuda finds no GPU or hangs. Import torch before anything loads "
            "libazgym_hip.so or another HIP library (INTEGRATION.md section 1).")


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
        _check_load_order()
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        both = mapped_hip_runtimes()
        if len(both) > 1:
            msg = (f"two HIP runtimes are mapped in this process ({', '.join(sorted(both))}). If torch was NOT imported first: import torch "
                   "before anything else loads a HIP library (INTEGRATION.md section 1). If it was (the wheel's libamdhip64 and the system's "
                   "differ in SONAME, or some other library maps a second copy on purpose) and both work side by side in your process: "
                   "set AZG_ALLOW_MULTI_HIP=1 to turn this error into a warning.")
            if os.environ.get("AZG_ALLOW_MULTI_HIP") == "1":
                import warnings
                warnings.warn(msg, RuntimeWarning)
            else:
                _lib = None
                raise HipRuntimeConflict(msg)
        _fns = _capi.bind(_lib, "azg_")
        ver = _fns["abi_version"]()
        if ver != _capi.ABI_VERSION:
            raise NativeLibraryMissing(f"ABI version mismatch: library {ver}, binding {_capi.ABI_VERSION}")
        _lib.azg_math_selftest.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t]
        _lib.azg_search_info.argtypes = [C.c_void_p, C.c_void_p]
    return _lib


def fns():
    lib()
    return _fns


class AzgSearchInfo(C.Structure):
    """include/azgym.h: azg_search_report (filled by azg_search_info)"""
    _fields_ = [("struct_size", C.c_int32), ("kernel_form", C.c_int32), ("tree_storage", C.c_int32), ("lds_exit", C.c_int32), ("spec", C.c_int32),
                ("waves", C.c_int32), ("groups", C.c_int32), ("tile_trees", C.c_int32),
                ("team_trees", C.c_int32), ("team_per_cu", C.c_int32), ("team_parts", C.c_int32), ("team_fallbacks", C.c_int32),
                ("max_records", C.c_int32), ("max_children", C.c_int32), ("last_ms", C.c_float), ("kernel_name", C.c_char * 192)]


KERNEL_FORMS = {-1: "none", 0: "persistent", 1: "per_layer", 2: "team"}
TREE_STORAGE = {0: "global", 1: "lds8", 2: "lds9"}
LDS_EXIT = {0: "resident", 1: "records", 2: "children", 3: "lds_size", 4: "forced", 5: "not_applicable"}


class HipEngine(_capi.Engine):
    """Batched MCTS engine on one MI355X."""

    def __init__(self, **kw):
        super().__init__(fns(), **kw)

    def search_info(self):
        """azg_search_info: what the last search ran as -- kernel form, where the trees lived and why, team-kernel fall-backs --
        as a dict (the enums as words: KERNEL_FORMS / TREE_STORAGE / LDS_EXIT)."""
        info = AzgSearchInfo()
        info.struct_size = C.sizeof(AzgSearchInfo)
        self._check(lib().azg_search_info(self._h, C.byref(info)))
        d = {k: getattr(info, k) for k, _ in AzgSearchInfo._fields_ if k not in ("struct_size", "kernel_name")}
        d["kernel_name"] = info.kernel_name.decode()
        d["kernel_form_id"] = info.kernel_form
        d["kernel_form"] = KERNEL_FORMS[info.kernel_form]
        d["tree_storage"] = TREE_STORAGE[info.tree_storage]
        d["lds_exit"] = LDS_EXIT[info.lds_exit]
        return d


def math_selftest(fn_id, x, device_id=0):
    import numpy as np

    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    rc = lib().azg_math_selftest(device_id, fn_id, _capi._ptr(x, C.c_double), _capi._ptr(out, C.c_double), x.size)
    if rc != 0:
        raise _capi.EngineError(rc, "azg_math_selftest failed")
    return out
