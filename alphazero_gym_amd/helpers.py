"""Host-side helpers with the reference's semantics (alphazero/helpers.py)."""
import pathlib
import random as _py_random

import numpy as np


def stable_normalizer(x: np.ndarray, temp: float) -> np.ndarray:
    """x[i]**temp / sum_i x[i]**temp, every entry divided by the largest one first so that the power cannot overflow
    (helpers.py:9-27).  Same operations in the same order as the reference: the sampled final action depends on the bits."""
    x = np.asarray(x)
    scaled = np.power(x / x.max(), temp)
    return np.abs(scaled / scaled.sum())


def argmax(x: np.ndarray) -> int:
    """Arg-max with a random tie-break drawn from Python's `random` (helpers.py:30-52).  The engine itself breaks ties by
    lowest index; the goldens assert that no tie occurs."""
    flat = np.ravel(x)
    if np.isnan(flat).any():
        print("argmax: the scores contain NaN, the result is meaningless")
    ties = np.flatnonzero(flat == flat.max())
    return _py_random.choice(ties)


def check_space(space):
    """(dimension tuple, is_discrete) of a gym-style space (helpers.py:55-78), duck-typed on `.n` / `.shape`."""
    n = getattr(space, "n", None)
    if n is not None:
        return (n,), True
    shape = getattr(space, "shape", None)
    if shape is not None:
        return tuple(shape), False
    raise NotImplementedError(f"unsupported kind of space: {type(space).__name__}")


def get_base_env(env):
    """The innermost environment under any stack of gym-style wrappers (helpers.py:95-99)."""
    return get_base_env(env.env) if hasattr(env, "env") else env


def store_actions(name: str, to_store: np.ndarray) -> None:
    """Dump the best action sequence to runs/<name>.npy (helpers.py:81-89)."""
    out_dir = pathlib.Path("runs")
    out_dir.mkdir(parents=True, exist_ok=True)
    np.save(out_dir / (name + ".npy"), to_store)
