#!/usr/bin/env python3
"""Determinism soak on the GPU box (python tools/soak.py [repeats]): the same searches over and over must give bit-identical
trees (catches races that the oracle comparison on small cases could miss)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

from alphazero_gym_amd import synthetic as O  # noqa: E402  (make_weights)
from alphazero_gym_amd import _capi, _native  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
CASES = [
    ("C 4096x200 2x256", dict(env_id=2, mode=1, n_trees=4096, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34), (3, [256, 256], 2, "elu")),
    ("C' 8232x200 2x256 (32-tree workgroups)", dict(env_id=2, mode=1, n_trees=8232, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=35), (3, [256, 256], 2, "elu")),
    ("B 4096x100 2x128", dict(env_id=0, mode=0, n_trees=4096, n_sims=100, c_uct=1.5, gamma=1.0, num_actions=2, seed=36), (4, [128, 128], 2, "relu")),
    ("E 1024x60 4x1024 (lock-step)", dict(env_id=2, mode=1, n_trees=1024, n_sims=60, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=37), (3, [1024] * 4, 2, "elu")),
    ("global trees 512x400 2x64", dict(env_id=2, mode=1, n_trees=512, n_sims=400, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=38), (3, [64, 64], 2, "elu")),
]


def digest(e):
    r, d = e.results(), e.dump_tree()
    h = hashlib.sha256()
    for k in sorted(r):
        h.update(np.ascontiguousarray(r[k]).tobytes())
    for k in sorted(d):
        h.update(np.ascontiguousarray(d[k]).tobytes())
    return h.hexdigest()


for name, kw, (ind, hidden, nd, act) in CASES:
    e = _native.HipEngine(**kw)
    e.set_weights(_capi.make_desc(ind, hidden, nd, act), O.make_weights(7, ind, hidden, nd))
    roots = e.synthetic_roots()
    ref = None
    reps = REPS if kw["n_trees"] * kw["n_sims"] < 2e6 else max(REPS // 4, 5)
    for i in range(reps):
        e.set_search_index(3)
        e.search(roots)
        dg = digest(e)
        if ref is None:
            ref = dg
        assert dg == ref, f"{name}: repetition {i} differs"
    print(f"{name}: {reps} repetitions identical ({ref[:16]})")
    e.close()
