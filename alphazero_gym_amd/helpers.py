"""Host-side helpers with the reference's semantics (alphazero/helpers.py)."""
import random
from pathlib import Path

import numpy as np


def stable_normalizer(x: np.ndarray, temp: float) -> np.ndarray:
    """x[i]**temp / sum_i x[i]**temp, scaled by the maximum first (helpers.py:9-27)."""
    x = (x / np.max(x)) ** temp
    return np.abs(x / np.sum(x))


def argmax(x: np.ndarray) -> int:
    """Arg-max with a random tie-break (helpers.py:30-52).  The engine itself breaks ties by lowest index."""
    x = x.flatten()
    if np.any(np.isnan(x)):
        print("Warning: Cannot argmax when vector contains nans, results will be wrong")
    winners = np.where(x == np.max(x))
    return random.choice(winners[0])


def check_space(space):
    """(dimension tuple, is_discrete) of a gym-style space (helpers.py:55-78), duck-typed on `.n` / `.shape`."""
    if hasattr(space, "n"):
        return (space.n,), True
    if hasattr(space, "shape"):
        return tuple(space.shape), False
    raise NotImplementedError("This type of space is not supported")


def get_base_env(env):
    """The innermost environment under any stack of gym-style wrappers (helpers.py:95-99)."""
    while hasattr(env, "env"):
        env = env.env
    return env


def store_actions(name: str, to_store: np.ndarray) -> None:
    """Dump the best action sequence to runs/<name>.npy (helpers.py:81-89)."""
    path = Path("runs/")
    path.mkdir(parents=True, exist_ok=True)
    np.save(path / f"{name}.npy", to_store)
