#!/usr/bin/env python3
"""Secondary measurements (not the headline bench.py line): BASELINE.json configs B and E, PCIe-inclusive search, self-play step.
GPU box only:  python tools/bench_configs.py [B|E|pcie|selfplay ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from alphazero_gym_amd import synthetic as O  # noqa: E402  (make_weights)
from alphazero_gym_amd import _capi, _native  # noqa: E402


def timed(eng, steps=5, warmup=1):
    for _ in range(warmup):
        eng.search_resident()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.search_resident()
    eng.sync()
    return (time.perf_counter() - t0) / steps


def config_B():
    e = _native.HipEngine(env_id=0, mode=0, n_trees=4096, n_sims=100, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)
    e.set_weights(_capi.make_desc(4, [128, 128], 2, "relu"), O.make_weights(34, 4, [128, 128], 2))
    e.upload_roots(e.synthetic_roots())
    dt = timed(e, 10, 2)
    print(f"config B  CartPole 4096 trees x 100 sims, 2x128 relu: {dt * 1e3:.3f} ms/search, {4096 * 100 / dt:.3e} sims/s")
    e.close()


def config_ref():
    """the reference's default continuous configuration (config/policy/ContinuousPolicy.yaml, config/mcts/MCTSContinuous.yaml):
    3x128 ELU trunk, 2-component Gaussian mixture head, 25 rollouts -- batched over 4096 games"""
    for ns in (25, 200):
        e = _native.HipEngine(env_id=2, mode=1, n_trees=4096, n_sims=ns, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
        e.set_weights(_capi.make_desc(3, [128, 128, 128], 6, "elu", num_components=2), O.make_weights(34, 3, [128, 128, 128], 6))
        e.upload_roots(e.synthetic_roots())
        dt = timed(e, 10, 2)
        print(f"reference default net (3x128 ELU, GMM-2), Pendulum 4096 trees x {ns} sims: {dt * 1e3:.3f} ms/search, {4096 * ns / dt:.3e} sims/s")
        e.close()


def config_E():
    B, NS = 1024, 200
    e = _native.HipEngine(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    e.set_weights(_capi.make_desc(3, [1024] * 4, 2, "elu"), O.make_weights(34, 3, [1024] * 4, 2))
    e.upload_roots(e.synthetic_roots())
    dt = timed(e, 2, 1)
    flop = 2 * (3 * 1024 + 3 * 1024 * 1024 + 1024 * 3)
    print(f"config E  Pendulum {B} trees/GPU x {NS} sims, 4x1024 elu: {dt * 1e3:.2f} ms/search, {B * NS / dt:.3e} sims/s, "
          f"{B * NS * flop / dt / 1e12:.1f} TFLOP/s = {100 * B * NS * flop / dt / 157.3e12:.1f} % of fp32 MFMA peak")
    e.close()


def pcie():
    e = _native.HipEngine(env_id=2, mode=1, n_trees=4096, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    e.set_weights(_capi.make_desc(3, [256, 256], 2, "elu"), O.make_weights(34, 3, [256, 256], 2))
    roots = e.synthetic_roots()
    e.search(roots); e.results()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        e.search(roots)      # host roots -> device, search, sync
        e.results()          # root statistics device -> host
    dt = (time.perf_counter() - t0) / n
    print(f"host-buffer round trip (azg_search + azg_results, config C): {dt * 1e3:.3f} ms/search, {4096 * 200 / dt:.3e} sims/s")
    e.close()


def selfplay():
    e = _native.HipEngine(env_id=2, mode=1, n_trees=4096, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    e.set_weights(_capi.make_desc(3, [256, 256], 2, "elu"), O.make_weights(34, 3, [256, 256], 2))
    e.selfplay_begin(200, False, 32)
    for _ in range(2):
        e.selfplay_step()
    e.sync(); e.selfplay_rows(clear=True)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        e.selfplay_step()
    e.sync()
    dt = (time.perf_counter() - t0) / n
    print(f"device-resident self-play step (search + action + env step + replay row, config C): {dt * 1e3:.3f} ms/step, "
          f"{4096 / dt:.3e} env steps/s, {4096 * 200 / dt:.3e} sims/s")
    e.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["B", "E", "pcie", "selfplay"]
    for w in which:
        {"B": config_B, "E": config_E, "pcie": pcie, "selfplay": selfplay, "ref": config_ref}[w]()
