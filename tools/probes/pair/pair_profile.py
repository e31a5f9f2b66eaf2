#!/usr/bin/env python3
"""Diagnostic: where the walker and server workgroups of the kernel pair (pair.cuh) spend their cycles (-DAZG_STAMPS build).
GPU box only:  make -C alphazero_gym_amd/csrc libazgym_hip_stamp.so && python tools/pair_profile.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AZG_HIP_LIB"] = os.path.join(ROOT, "alphazero_gym_amd", "csrc", os.environ.get("PAIR_LIB", "libazgym_hip_stamp.so"))
os.environ.setdefault("AZG_PAIR", "2")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alphazero_gym_amd import _capi, _native  # noqa: E402
from alphazero_gym_amd.synthetic import make_weights  # noqa: E402

B, NS = int(os.environ.get("C_TREES", "8192")), 200
e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
e.set_weights(_capi.make_desc(3, [256, 256], 2, "elu"), make_weights(34, 3, [256, 256], 2))
e.upload_roots(e.synthetic_roots())
for _ in range(2):
    e.search_resident()
e.sync()
print("kernel ms", e.last_search_ms())
pairs = min((B + 31) // 32, 256)
rows = pairs * 4
buf = np.zeros((rows, 16), np.uint64)
lib = _native.lib()
lib.azg_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
assert lib.azg_debug_stamps(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), rows) >= rows
print("kernel form", lib.azg_debug_kernel_form(e._h))
w = buf.astype(np.float64) / (2 * (NS + 1))     # cycles per half-step (one group's evaluation / walk)
names = ["walker: wait for results", "walker: fetch results", "walker: finish leaf + backup", "walker: select / step / expand", "walker: hand in",
         "walker: whole loop", "walker: loop body incl. swap", "walker: fast link (1 = yes)", "server: wait for observations", "server: fetch + network", "server: store + arrive", "server: whole loop"]
for i, nm in enumerate(names):
    if not nm:
        continue
    v = w[:, i] if i < 8 else w[0::4, i]
    print(f"  {nm:34s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f} cycles per half-step")
