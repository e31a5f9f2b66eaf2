#!/usr/bin/env python3
"""Config E (Pendulum, 1024 trees/GPU x 200 sims, 4x1024 ELU) on the lock-step path under its switches: persistent team kernel,
pipelines, fused first layer.  GPU box only:  python tools/sweep_e.py [trees]"""
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alphazero_gym_amd import _capi, _native  # noqa: E402
from alphazero_gym_amd.synthetic import make_weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NS = 200
flop = 2 * (3 * 1024 + 3 * 1024 * 1024 + 1024 * 3)
ref = None
for pipes, team, fuse in [(1, 0, 0), (1, 1, 0), (1, 0, 0), (1, 1, 0)]:
    os.environ.update(AZG_LS_PIPES=str(pipes), AZG_LS_TEAM=str(team), AZG_LS_FUSE0=str(fuse))
    e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    e.set_weights(_capi.make_desc(3, [1024] * 4, 2, "elu"), make_weights(34, 3, [1024] * 4, 2))
    e.upload_roots(e.synthetic_roots())
    e.search_resident(); e.sync()
    ms = []
    t0 = time.perf_counter()
    for _ in range(4):
        e.search_resident()
        ms.append(e.last_search_ms())
    wall = (time.perf_counter() - t0) / 4 * 1e3
    r = e.results()
    key = (r["counts"].tobytes(), r["Q"].tobytes())
    if ref is None:
        ref = key
    m = float(np.median(ms))
    print(f"pipes {pipes} team {team} fuse0 {fuse}: {m:7.3f} ms/search (wall {wall:7.3f}), {m * 1e3 / (NS + 1):6.1f} us/step, "
          f"{B * NS * flop / (m * 1e-3) / 157.3e12 * 100:5.1f} % of fp32 MFMA peak, identical results: {key == ref}", flush=True)
    e.close()
