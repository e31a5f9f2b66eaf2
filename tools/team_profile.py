#!/usr/bin/env python3
"""Diagnostic: where the workgroups of the persistent team kernel (config E) spend their cycles (-DAZG_STAMPS build).
GPU box only:  make -C alphazero_gym_amd/csrc " + os.environ.get("TEAM_LIB", "libazgym_hip_stamp.so") + " && python tools/team_profile.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AZG_HIP_LIB"] = os.path.join(ROOT, "alphazero_gym_amd", "csrc", "" + os.environ.get("TEAM_LIB", "libazgym_hip_stamp.so") + "")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alphazero_gym_amd import _capi, _native  # noqa: E402
from alphazero_gym_amd.synthetic import make_weights  # noqa: E402

B, NS = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 200
e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
e.set_weights(_capi.make_desc(3, [1024] * 4, 2, "elu"), make_weights(34, 3, [1024] * 4, 2))
e.upload_roots(e.synthetic_roots())
for _ in range(2):
    e.search_resident()
e.sync()
print("kernel ms", e.last_search_ms())
rows = max((B + 3) // 4 * 4, (B + 31) // 32 * 24, 16)       # azg_engine.hip stamp_rows
buf = np.zeros((rows, 16), np.uint64)
lib = _native.lib()
lib.azg_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
assert lib.azg_debug_stamps(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), rows) == rows
flat = buf.reshape(-1).astype(np.float64) / (NS + 1)
names = ["wait for observations", "tile layer 1 (+ layer 0)", "tile layer 2", "tile layer 3", "arrive + wait between layers",
         "wait for the last layer", "tree phases", "whole loop"]
kname = e.search_info()["kernel_name"]
TT = 64 if kname.rstrip(">").endswith(" 64") else 32
n_wg = (B + TT - 1) // TT * 16                              # teams of TT trees x 16 workgroups (HP = 1024)
w = flat[:n_wg * 8].reshape(n_wg, 8)                        # [workgroup = team * 16 + slice][slot], cycles per step
parts = flat[n_wg * 8:n_wg * 24].reshape(n_wg, 16)          # the tree phases' own parts
print("kernel", kname, "workgroups", n_wg)
for i, nm in enumerate(names):
    v = w[:, i]
    print(f"  {nm:30s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f} cycles/step")
print("per step at 2.4 GHz:", w[:, 7].mean() / 2400, "us")
part_names = ["head partials into LDS", "tree phase A", "tree phase B", "first layer + arrive", "A: finish leaf (stampa build)", "A: backup (stampa build)",
              "A: re-scoring + resume (stampa build)", "B: a level's selection", "B: per level scores + arg-max", "B: per level chosen record",
              "B: per level path slot + cold prefetch", "B: all levels", "(levels)", "B: whole descent", "B: widening", "B: env step + observation"]
print("tree phases by part (thread 0's clock; 'per level' rows are sums over the step's levels):")
for i, nm in enumerate(part_names):
    v = parts[:, i]
    print(f"  {nm:42s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f}" + (" levels/step" if i == 12 else " cycles/step"))
