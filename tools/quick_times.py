#!/usr/bin/env python3
"""Kernel milliseconds (HIP events on the engine stream, median of 9 after 3 warm-up searches) of the BASELINE shapes, for A/B runs
on the GPU box:  python tools/quick_times.py [C C8192 B E B8192 ...]   (AZG_HIP_LIB selects another build of the library)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alphazero_gym_amd import _capi, _native  # noqa: E402
from alphazero_gym_amd.synthetic import make_weights  # noqa: E402

PEND = dict(env_id=2, mode=1, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
CART = dict(env_id=0, mode=0, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)
MCC = dict(env_id=4, mode=1, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34, action_bound=1.0)   # MountainCarContinuous-v0
SHAPES = {
    "C": (PEND, 4096, 200, 3, [256, 256], "elu"), "C8192": (PEND, 8192, 200, 3, [256, 256], "elu"), "C16384": (PEND, 16384, 200, 3, [256, 256], "elu"),
    "B": (CART, 4096, 100, 4, [128, 128], "relu"), "B2048": (CART, 2048, 100, 4, [128, 128], "relu"), "B1024": (CART, 1024, 100, 4, [128, 128], "relu"),
    "B16384": (CART, 16384, 100, 4, [128, 128], "relu"), "P128": (PEND, 4096, 200, 3, [128, 128], "elu"), "B8192": (CART, 8192, 100, 4, [128, 128], "relu"), "B256": (CART, 8192, 100, 4, [256, 256], "relu"),
    "M": (MCC, 4096, 200, 2, [256, 256], "elu"), "M8192": (MCC, 8192, 200, 2, [256, 256], "elu"),
    "E": (PEND, 1024, 200, 3, [1024] * 4, "elu"), "E2048": (PEND, 2048, 200, 3, [1024] * 4, "elu"), "E3072": (PEND, 3072, 200, 3, [1024] * 4, "elu"),
}


def main():
    for name in (sys.argv[1:] or ["C", "C8192", "B"]):
        if name.startswith("X"):   # X<trees>: config E's network at that many trees
            SHAPES[name] = (PEND, int(name[1:]), 200, 3, [1024] * 4, "elu")
        kw, B, ns, ind, hidden, act = SHAPES[name]
        e = _native.HipEngine(n_trees=B, n_sims=ns, **kw)
        e.set_weights(_capi.make_desc(ind, hidden, 2, act), make_weights(34, ind, hidden, 2))
        e.upload_roots(e.synthetic_roots())
        for _ in range(3):
            e.search_resident()
        e.sync()
        ms = []
        for _ in range(9):
            e.search_resident()
            ms.append(e.last_search_ms())
        kname = e.search_info()["kernel_name"]
        r = e.results()
        assert (r["counts"].sum(1) == ns).all()
        print(f"{name:7s} {np.median(ms):8.4f} ms (min {min(ms):.4f})  {B * ns / np.median(ms) / 1e3:10.4e} sims/s  {kname}", flush=True)
        e.close()


if __name__ == "__main__":
    main()
