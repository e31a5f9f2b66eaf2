#!/usr/bin/env python3
"""GPU box: the team kernel's forms (32-tree teams: two, three workgroups per CU; 64-tree teams: two, three) 40 searches each: identical results every time, no
fall-back to the per-layer launches (python tools/team_soak.py)."""
import sys, hashlib, ctypes as C
sys.path.insert(0, ".")
import numpy as np
from alphazero_gym_amd import _capi, _native
from alphazero_gym_amd.synthetic import make_weights
for B in (1024, 1536, 2048, 3072, 4096):
    e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    e.set_weights(_capi.make_desc(3, [1024] * 4, 2, "elu"), make_weights(34, 3, [1024] * 4, 2))
    e.upload_roots(e.synthetic_roots())
    hs = set()
    for i in range(40):
        e.set_search_index(0)
        e.search_resident(); e.sync()
        r = e.results()
        hs.add(hashlib.sha1(r["counts"].tobytes() + r["Q"].tobytes()).hexdigest())
    fb = e.search_info()["team_fallbacks"]
    print(B, "distinct results:", len(hs), "fallbacks:", fb, "ms", e.last_search_ms())
    e.close()
