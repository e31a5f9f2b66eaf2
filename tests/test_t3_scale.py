"""Tier T3 at BASELINE scale (VERDICT r03 row g): visit-count parity with the reference run END TO END -- the reference's
MCTSContinuous.search / MCTSDiscrete.search (alphazero/search/mcts.py:418-462, 656-702) with its REAL torch policies
(alphazero/network/policies.py:340-352, 436-499) -- on the engine's own synthetic roots of configs C (all 4096 trees, 2x256 ELU,
200 rollouts), B (all 4096 trees, CartPole 2x128 ReLU, 100 rollouts) and E (all 1024 trees of the per-GPU leg, 4x1024 ELU, 200 rollouts), and -- 1024 trees
each -- of the reference's own DEFAULT configurations (config/mcts/*.yaml, config/policy/*.yaml): the 2-component mixture head on a
3x128 ELU trunk with 25 rollouts, and CartPole with 8 rollouts and epsilon-greedy 0.1 (the engine's draws injected as `random`);
and 1024 trees of gym MountainCar-v0 (three actions) with the reference's DiscretePolicy; and 1024 trees of gym
MountainCarContinuous-v0 -- the continuous search over an env whose episodes end (terminal nodes: mcts.py:619-623, 682) -- with the
reference's DiagonalNormalPolicy; and 1024 trees of gym Acrobot-v1 (six observations) with its DiscretePolicy:
tests/golden/t3_scale.npz, written by tests/golden/gen_golden.py `scale` from the imported reference.

The networks differ from torch's by ~1e-7 (summation order), so a selection whose two best scores are closer than that could
legitimately flip.  The test therefore reports a MATCH RATE and attributes every mismatching tree: the fixture holds, for the trees
whose counts differed from the oracle's when it was generated, the reference's per-trace leaf records; the test finds the first
trace at which the oracle leaves the reference's sequence and requires the oracle's tightest arg-max gap on that trace to be below
1e-6 (a near-tie) -- anything else is a real difference and fails.  (When this fixture was generated: 1 mismatching tree of 11392 -- config C's tree 1815 leaves the reference's sequence at trace 24,
where the oracle's two best scores are 2.07e-8 apart.)
"""
import os

import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from alphazero_gym_amd import _capi

TOL = 1e-5          # north_star: Q-values / policy outputs within 1e-5
NEAR_TIE = 1e-6     # a selection whose best two scores are closer than this may flip under a ~1e-7 network difference

LEGS = {   # tag: (engine kwargs, in_dim, hidden, activation, n_sims, network outputs besides the value, mixture components, weight seed)
    "c": (dict(env_id=2, mode=1, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34), 3, [256, 256], "elu", 200, 2, 0, 34),
    "b": (dict(env_id=0, mode=0, c_uct=1.5, gamma=1.0, num_actions=2, seed=34), 4, [128, 128], "relu", 100, 2, 0, 34),
    "e": (dict(env_id=2, mode=1, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34), 3, [1024] * 4, "elu", 200, 2, 0, 34),
    # the reference's own default configurations (config/mcts/*.yaml, config/policy/*.yaml): mixture head, 25 rollouts; epsilon-greedy 0.1, 8 rollouts
    "g": (dict(env_id=2, mode=1, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34), 3, [128, 128, 128], "elu", 25, 6, 2, 35),
    "d": (dict(env_id=0, mode=0, c_uct=1.5, gamma=1.0, num_actions=2, epsilon=0.1, seed=34), 4, [128, 128], "relu", 8, 2, 0, 34),
    # three actions end to end: gym MountainCar-v0 with the reference's DiscretePolicy (2x64 ReLU), 60 rollouts, gamma 0.99
    "m": (dict(env_id=3, mode=0, c_uct=0.8, gamma=0.99, num_actions=3, seed=34), 2, [64, 64], "relu", 60, 3, 0, 34),
    # the continuous search over an env whose episodes END (VERDICT r04 row h; mcts.py:619-623, 682): gym MountainCarContinuous-v0 with
    # the reference's DiagonalNormalPolicy (2x256 ELU, action bound 1), 120 rollouts, roots on the slope below the flag
    "h": (dict(env_id=4, mode=1, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, action_bound=1.0, seed=34), 2, [256, 256], "elu", 120, 2, 0, 34),
    # six observations end to end: gym Acrobot-v1 with the reference's DiscretePolicy (2x64 ReLU), 50 rollouts, roots on the upswing
    "a": (dict(env_id=5, mode=0, c_uct=0.8, gamma=0.99, num_actions=3, seed=34), 6, [64, 64], "relu", 50, 3, 0, 34),
}
N_TREES_ALL = 4096 + 4096 + 1024 + 1024 + 1024 + 1024 + 1024 + 1024


def acrobot_scale_roots(synthetic):
    """Leg a's roots (the same formula as tests/golden/gen_golden.py acrobot_scale_roots): the engine's synthetic Acrobot roots (hanging,
    at rest) mapped onto the upswing -- theta1 = 1.4 + 6 s0, theta2 = 8 s1, dtheta1 = 3 + 20 s2, dtheta2 = 20 s3; a mapped root that is
    already above the line is replaced by a fixed one."""
    from alphazero_gym_amd.envs import AcrobotEnv
    s = np.asarray(synthetic)
    roots = np.stack([1.4 + 6.0 * s[:, 0], 8.0 * s[:, 1], 3.0 + 20.0 * s[:, 2], 20.0 * s[:, 3]], 1)
    for i, r in enumerate(roots):
        if AcrobotEnv(state=r)._terminal():
            roots[i] = [1.0, 0.0, 0.5, 0.0]
    return roots


def mcc_scale_roots(synthetic):
    """Leg h's roots (the same formula as tests/golden/gen_golden.py mcc_scale_roots): the engine's synthetic MountainCar roots (valley,
    at rest) mapped onto the slope below the flag, u = (position + 0.6) / 0.2 -> position 0.25 + 0.199 u, velocity 0.02 + 0.05 frac(17 u)."""
    u = (np.asarray(synthetic)[:, 0] + 0.6) / 0.2
    return np.stack([0.25 + 0.199 * u, 0.02 + 0.05 * ((17.0 * u) % 1.0)], 1)

NAMES = {"c": "config C", "b": "config B", "e": "config E", "g": "the reference's default continuous setup (mixture head, 25 rollouts)",
         "d": "the reference's default discrete setup (epsilon-greedy 0.1, 8 rollouts)",
         "m": "MountainCar-v0 (three actions, 60 rollouts)",
         "h": "MountainCarContinuous-v0 (continuous search with terminal nodes, 120 rollouts)",
         "a": "Acrobot-v1 (six observations, three actions, 50 rollouts)"}


def _network(leg):
    kw, in_dim, hidden, act, n_sims, n_dist, ncomp, wseed = leg
    return _capi.make_desc(in_dim, hidden, n_dist, act, num_components=ncomp), O.make_weights(wseed, in_dim, hidden, n_dist)


def first_divergence(ref_leaf, own_leaf):
    """Index of the first trace whose final record differs, or -1."""
    # (entries: the trace's final record | its parent node's record << 16 -- together they pin the trace's path)
    d = np.nonzero(np.asarray(ref_leaf) != np.asarray(own_leaf))[0]
    return int(d[0]) if d.size else -1


def attribute(tag, tree, ref_leaf, root):
    """Re-search one tree on the oracle with trace diagnostics; (first diverging trace, the oracle's tightest arg-max gap on it)."""
    kw, n_sims = LEGS[tag][0], LEGS[tag][4]
    o = O.OracleEngine(n_trees=1, n_sims=n_sims, tree_id_base=int(tree), **kw)
    o.set_weights(*_network(LEGS[tag]))
    o.trace_enable()
    o.search(root[None, :])
    leaf, margin = o.trace_get()
    o.close()
    d = first_divergence(ref_leaf, leaf[0])
    return d, (float(margin[0, d]) if d >= 0 else float("inf"))


def _scale(engine_cls, legs=("c", "b", "e", "g", "d", "m", "h", "a")):
    z = np.load(os.path.join(P.GOLDEN, "t3_scale.npz"))
    total = matched = 0
    lines = []
    for tag in legs:
        kw, n_sims = LEGS[tag][0], LEGS[tag][4]
        roots = z[f"{tag}_roots"]
        B = len(roots)
        e = engine_cls(n_trees=B, n_sims=n_sims, **kw)
        # the fixture's roots are the engine's own synthetic roots 0..B-1 (leg h: mapped onto the slope below the flag)
        own = e.synthetic_roots()
        np.testing.assert_array_equal(mcc_scale_roots(own) if tag == "h" else (acrobot_scale_roots(own) if tag == "a" else own), roots)
        e.set_weights(*_network(LEGS[tag]))
        e.search(roots)
        r = e.results()
        if tag == "h":
            # the leg is about terminal nodes: most trees must contain some, and traces that ended in an EXISTING terminal node
            # created no record (Pendulum: n_records == n_sims + 1 always)
            d = e.dump_tree()
            with_terminal = ((d["node_flags"] & 2) != 0).any(1)
            short = d["n_records"] < n_sims + 1
            assert with_terminal.mean() > 0.5 and short.mean() > 0.3, (with_terminal.mean(), short.mean())
            lines.append(f"   leg h: {int(with_terminal.sum())} of {B} trees hold terminal nodes, {int(short.sum())} ran traces that ended in an "
                         f"existing terminal node ({int((n_sims + 1 - d['n_records']).sum())} such traces in all)")
        e.close()
        K = z[f"{tag}_counts"].shape[1]
        ref_counts = z[f"{tag}_counts"].astype(np.int32)
        same = (r["n_children"] == z[f"{tag}_n_children"]) & (r["counts"][:, :K] == ref_counts).all(1)
        bad = np.nonzero(~same)[0]
        known = {int(t): i for i, t in enumerate(z[f"{tag}_mismatch_ids"])}
        for t in bad:
            # a tree that differs from the reference must be one the generator saw differing on the oracle too (HIP == oracle bit
            # for bit), and its first diverging trace must sit on a near-tie of the oracle's scores
            assert int(t) in known, f"{tag} tree {t}: visit counts differ from the reference's: {r['counts'][t][:K]} vs {ref_counts[t]}"
            d, gap = attribute(tag, t, z[f"{tag}_mismatch_ref_leaf"][known[int(t)]], roots[t])
            assert d >= 0 and gap < NEAR_TIE, f"{tag} tree {t}: first diverging trace {d} with an arg-max gap of {gap:.3e}: not a near-tie"
            lines.append(f"   {tag} tree {t}: leaves the reference at trace {d}, arg-max gap {gap:.3e} (near-tie)")
        # Q / actions / value target of the stored trees that match
        F = len(z[f"{tag}_Q"])
        ok = np.nonzero(same[:F])[0]
        nch = r["n_children"][:F]
        mask = (np.arange(K)[None, :] < nch[:, None])[ok]
        np.testing.assert_allclose(np.where(mask, r["Q"][:F, :K][ok], 0.0), np.where(mask, z[f"{tag}_Q"][ok], 0.0), atol=TOL, rtol=TOL)
        np.testing.assert_allclose(r["v_target"][:F][ok], z[f"{tag}_v_target"][ok], atol=TOL, rtol=TOL)
        if f"{tag}_actions" in z.files:
            np.testing.assert_allclose(np.where(mask, r["actions"][:F, :K][ok], 0.0), np.where(mask, z[f"{tag}_actions"][ok], 0.0), atol=TOL, rtol=TOL)
        # the selected action index (max visit count, first index: agents.py:524-527) follows from identical counts
        total += B
        matched += int(same.sum())
        lines.append(f"T3 scale, {NAMES[tag]}: {int(same.sum())} of {B} trees have the reference's visit counts "
                     f"(match rate {same.mean():.4f}); Q / actions / value target within {TOL} on the {len(ok)} stored trees")
    lines.append(f"T3 scale, all legs: visit-count match rate {matched}/{total} = {matched / total:.4f} ({engine_cls.__name__})")
    print("\n" + "\n".join(lines))
    return matched, total


def test_oracle_matches_reference_with_torch_policy_at_baseline_scale():
    matched, total = _scale(O.OracleEngine)
    assert total == N_TREES_ALL
    assert matched / total >= 0.999    # (every mismatch has already been attributed to a near-tie above)


@pytest.mark.gpu
def test_hip_matches_reference_with_torch_policy_at_baseline_scale(capsys):
    from alphazero_gym_amd import _native
    _native.lib()
    matched, total = _scale(_native.HipEngine)
    assert total == N_TREES_ALL
    assert matched / total >= 0.999
    with capsys.disabled():   # the rate belongs in the GPU run's log
        print(f"\n[T3 scale on HIP] visit-count match rate vs the reference with its torch policies: {matched}/{total} = {matched / total:.4f}")


def test_first_divergence_and_trace_diagnostics():
    """The attribution machinery itself: two oracle searches whose networks differ slightly leave each other's trace sequence at a
    definite trace; before it every leaf is identical, and the reported gap is the smallest arg-max gap met on that trace."""
    assert first_divergence([1, 2, 3], [1, 2, 3]) == -1
    assert first_divergence([1, 2, 3], [1, 5, 3]) == 1
    kw, in_dim, hidden, act, n_sims = LEGS["c"][:5]
    w = O.make_weights(34, in_dim, hidden, 2)
    leaves = []
    for scale in (1.0, 1.0 + 2e-3):
        o = O.OracleEngine(n_trees=4, n_sims=n_sims, **kw)
        o.set_weights(_capi.make_desc(in_dim, hidden, 2, act), (w * np.float32(scale)).astype(np.float32))
        o.trace_enable()
        o.search(o.synthetic_roots())
        leaf, margin = o.trace_get()
        o.close()
        leaves.append(leaf)
        assert leaf.shape == (4, n_sims)
        np.testing.assert_array_equal(leaf & 0xffff, np.arange(1, n_sims + 1)[None, :].repeat(4, 0))   # Pendulum: every trace ends in a new record
        assert ((leaf >> 16) < (leaf & 0xffff)).all()                       # ... whose parent node is an older record
        assert (margin > 0).all()                                         # no exact tie on these roots
        assert np.isinf(margin).any() and np.isfinite(margin).any()       # traces that only widened the root score nothing
    ds = [first_divergence(leaves[0][t], leaves[1][t]) for t in range(4)]
    assert any(d > 0 for d in ds)
    for t, d in enumerate(ds):
        if d > 0:
            np.testing.assert_array_equal(leaves[0][t][:d], leaves[1][t][:d])
