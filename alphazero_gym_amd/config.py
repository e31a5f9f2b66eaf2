"""Minimal stand-in for hydra's `_target_` instantiation (the reference builds its agents with hydra.utils.call /
hydra.utils.instantiate, agents.py:80-86; hydra is an experiment-management dependency, not part of the hot path).

A config is a plain mapping: {"_target_": "pkg.mod.Class", **kwargs}.  Targets that point at the reference's module
paths (``alphazero.…``) are mapped onto this package, so the reference's YAML files work unchanged once loaded."""
import importlib
from typing import Any, Mapping

_ALIASES = {
    "alphazero.": "alphazero_gym_amd.",
}


def resolve(target: str):
    for old, new in _ALIASES.items():
        if target.startswith(old):
            target = new + target[len(old):]
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(cfg: Any, **overrides) -> Any:
    """hydra.utils.instantiate / hydra.utils.call for flat configs; objects that are not configs pass through."""
    if not isinstance(cfg, Mapping) or "_target_" not in cfg:
        if overrides and callable(cfg):
            return cfg(**overrides)
        return cfg
    kwargs = {k: v for k, v in cfg.items() if k != "_target_"}
    kwargs.update(overrides)
    return resolve(cfg["_target_"])(**kwargs)
