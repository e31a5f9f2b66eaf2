#!/usr/bin/env python3
"""One line per call: config C's network at C_TREES trees (default 8192), ms per search, which kernel form ran, a digest of the
results.  AZG_PAIR=0|1|2 selects the one-kernel form / the walker + server pair (pair.cuh)."""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alphazero_gym_amd import _capi, _native  # noqa: E402
from alphazero_gym_amd.synthetic import make_weights  # noqa: E402

B, NS = int(os.environ.get("C_TREES", "8192")), int(os.environ.get("C_SIMS", "200"))
env = int(os.environ.get("C_ENV", "2"))
if env == 0:
    e = _native.HipEngine(env_id=0, mode=0, n_trees=B, n_sims=NS, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)
    hidden, in_dim, act = [128, 128], 4, "relu"
else:
    e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    hidden, in_dim, act = [256, 256], 3, "elu"
e.set_weights(_capi.make_desc(in_dim, hidden, 2, act), make_weights(34, in_dim, hidden, 2))
e.upload_roots(e.synthetic_roots())
e.search_resident(); e.sync()
ms = []
for _ in range(8):
    e.search_resident()
    ms.append(e.last_search_ms())
r = e.results()
lib = _native.lib()
form = lib.azg_debug_kernel_form(C.c_void_p(e._h.value))
fb = lib.azg_debug_team_fallbacks(C.c_void_p(e._h.value))
flop = 2 * (in_dim * hidden[0] + hidden[0] * hidden[1] + hidden[1] * 3)
m = float(np.median(ms))
print(f"AZG_PAIR={os.environ.get('AZG_PAIR', '-')} trees {B}: {m:.3f} ms/search (min {min(ms):.3f}), {B * NS / (m * 1e-3):.3e} sims/s, "
      f"{B * NS * flop / (m * 1e-3) / 157.3e12 * 100:.1f} % of peak, form {form}, fallbacks {fb}, "
      f"results {hashlib.md5(r['counts'].tobytes() + r['Q'].tobytes()).hexdigest()[:8]}")
e.close()
