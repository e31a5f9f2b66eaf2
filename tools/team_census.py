#!/usr/bin/env python3
"""Diagnostic (-DAZG_STAMPS -DTEAM_CENSUS build): which CU every workgroup of the team kernel ran on."""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AZG_HIP_LIB"] = os.path.join(ROOT, "alphazero_gym_amd", "csrc", "libazgym_hip_x_census.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alphazero_gym_amd import _capi, _native  # noqa: E402
from alphazero_gym_amd.synthetic import make_weights  # noqa: E402

B, NS = 1024, 20
e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
e.set_weights(_capi.make_desc(3, [1024] * 4, 2, "elu"), make_weights(34, 3, [1024] * 4, 2))
e.upload_roots(e.synthetic_roots())
e.search_resident(); e.sync()
rows = (B + 15) // 16 * 4
buf = np.zeros((rows, 16), np.uint64)
lib = _native.lib()
lib.azg_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
assert lib.azg_debug_stamps(e._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), rows) == rows
w = buf.reshape(-1, 8)[:, 0]
place = collections.defaultdict(list)
for wg, v in enumerate(w):
    v = int(v)
    hw, xcc, blk = v & 0xffffffff, (v >> 32) & 0xf, v >> 40
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    place[(xcc, se, sh, cu)].append((blk, wg // 16, wg % 16))
print("distinct CUs:", len(place), "workgroups per CU:", collections.Counter(len(v) for v in place.values()))
for k in sorted(place)[:12]:
    print("xcc %d se %d sh %d cu %2d:" % k, " ".join("blk %3d (team %2d slice %2d)" % t for t in sorted(place[k])))
same = sum(1 for v in place.values() if len(v) == 2 and v[0][1] == v[1][1])
print("CUs whose two workgroups are of the same team:", same)
blk_xcc = collections.defaultdict(set)
for k, v in place.items():
    for t in v:
        blk_xcc[t[0] % 8].add(k[0])
print("blockIdx % 8 -> xcc:", {k: sorted(v) for k, v in sorted(blk_xcc.items())})
