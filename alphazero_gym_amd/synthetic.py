"""Deterministic synthetic network weights (benchmarks, smoke test, fixtures): numpy only.

``make_weights`` draws a policy/value MLP in torch's ``nn.Linear`` default-init ranges (U(-1/sqrt(fan_in), 1/sqrt(fan_in)))
from numpy's PCG64, in the blob layout of include/azgym.h (state_dict order), so that a seed identifies the weights."""
import numpy as np


def make_weights(seed, in_dim, hidden, n_dist, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    k = in_dim
    for h in list(hidden):
        b = 1.0 / np.sqrt(k)
        parts.append(rng.uniform(-b, b, size=(h, k)).astype(np.float32).ravel() * np.float32(scale))
        parts.append(rng.uniform(-b, b, size=(h,)).astype(np.float32))
        k = h
    b = 1.0 / np.sqrt(k)
    parts.append(rng.uniform(-b, b, size=(1, k)).astype(np.float32).ravel())
    parts.append(rng.uniform(-b, b, size=(1,)).astype(np.float32))
    parts.append(rng.uniform(-b, b, size=(n_dist, k)).astype(np.float32).ravel() * np.float32(scale))
    parts.append(rng.uniform(-b, b, size=(n_dist,)).astype(np.float32))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def add_layernorm(blob, in_dim, hidden, n_dist, seed):
    """Insert LayerNorm weight/bias after every trunk layer of a make_weights blob: per layer W, b, ln_w, ln_b."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts, p, k = [], 0, in_dim
    for h in hidden:
        parts.append(blob[p:p + h * k + h]); p += h * k + h
        parts.append(rng.uniform(0.5, 1.5, (h,)).astype(np.float32))
        parts.append(rng.uniform(-0.3, 0.3, (h,)).astype(np.float32))
        k = h
    parts.append(blob[p:])
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)
