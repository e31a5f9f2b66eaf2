#!/usr/bin/env python3
"""Size sweep on the GPU box (python tools/stress.py): very large batches, long searches (trees in global memory), wide nets,
single-tree and single-simulation corner cases; checks the count invariant and prints throughput."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from alphazero_gym_amd import synthetic as O
from alphazero_gym_amd import _capi, _native
def run(B, NS, hidden, env=2, mode=1, **kw):
    base = dict(env_id=env, mode=mode, n_trees=B, n_sims=NS, seed=5)
    base.update(dict(c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5) if mode == 1 else dict(c_uct=1.5, gamma=1.0, num_actions=2))
    base.update(kw)
    e = _native.HipEngine(**base)
    ind, nd = (3, 2) if mode == 1 else (4, 2)
    e.set_weights(_capi.make_desc(ind, hidden, nd, "elu"), O.make_weights(3, ind, hidden, nd))
    roots = e.synthetic_roots()
    t = time.time(); e.search(roots); r = e.results(); dt = time.time() - t
    assert (r["counts"].sum(1) == NS).all(), "count sum"
    assert np.isfinite(r["Q"][r["counts"] > 0]).all()
    print(f"B={B} NS={NS} hidden={hidden} env={env}: {dt*1e3:.1f} ms, {B*NS/dt:.3e} sims/s, kernel {e.last_search_ms():.2f} ms, kmax {r['n_children'].max()}")
    e.close()
run(65536, 200, [256, 256])
run(100000, 50, [128, 128])
run(256, 2000, [64, 64])
run(64, 5000, [64, 64], c_pw=3.0, kappa=0.6)
run(4096, 1000, [128, 128], env=0, mode=0)
run(1, 200, [256, 256])
run(17, 1, [64])
run(3000, 200, [1024, 1024, 1024])
