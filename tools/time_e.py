#!/usr/bin/env python3
"""One line: config E (Pendulum, 1024 trees x 200 sims, 4x1024 ELU) ms per search of the library in AZG_HIP_LIB (default: product)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from alphazero_gym_amd import _capi, _native  # noqa: E402
from alphazero_gym_amd.synthetic import make_weights  # noqa: E402

B, NS = int(os.environ.get("E_TREES", "1024")), 200
e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
e.set_weights(_capi.make_desc(3, [1024] * 4, 2, "elu"), make_weights(34, 3, [1024] * 4, 2))
e.upload_roots(e.synthetic_roots())
e.search_resident(); e.sync()
ms = []
for _ in range(6):
    e.search_resident()
    ms.append(e.last_search_ms())
r = e.results()
import hashlib
flop = 2 * (3 * 1024 + 3 * 1024 * 1024 + 1024 * 3)
m = float(np.median(ms))
print(f"{os.path.basename(os.environ.get('AZG_HIP_LIB', 'product'))}: {m:.3f} ms/search, {m * 1e3 / (NS + 1):.1f} us/step, "
      f"{B * NS * flop / (m * 1e-3) / 157.3e12 * 100:.1f} % of peak, results {hashlib.md5(r['counts'].tobytes() + r['Q'].tobytes()).hexdigest()[:8]}")
e.close()
