"""GPU (-m gpu): the HIP engine against the CPU oracle and the reference's golden vectors, through the C ABI."""
import os

import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from alphazero_gym_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from alphazero_gym_amd import _native
    _native.lib()
    return _native


MATH_INPUTS = {
    0: np.linspace(-90, 10, 20001), 1: np.linspace(-20, 3, 20001), 2: np.linspace(-12, 12, 20001),
    3: np.linspace(1e-8, 1.0, 20001), 4: np.linspace(0, 0.99999, 20001), 5: np.linspace(-100, 100, 20001),
    6: np.linspace(-100, 100, 20001), 7: np.linspace(-100, 100, 20001), 8: np.arange(20001, dtype=np.float64),
    9: np.linspace(-50, 50, 20001), 10: np.linspace(0, 50, 20001), 11: np.linspace(-50, 50, 20001),
    12: np.concatenate([np.arange(1, 20001, dtype=np.float64), np.linspace(2e4, 2e9, 20001)]),   # sqrt(n + 1) beyond the host table
}


@pytest.mark.parametrize("fn_id", sorted(MATH_INPUTS))
def test_device_math_is_bit_identical_to_host(native, fn_id):
    x = MATH_INPUTS[fn_id]
    host = O.math_eval(fn_id, x)
    dev = native.math_selftest(fn_id, x)
    np.testing.assert_array_equal(host.view(np.uint64), dev.view(np.uint64))


def test_mfma_f32_is_a_k_ordered_fma_chain(native):
    """v_mfma_f32_16x16x4_f32 must accumulate D = fma(a_k, b_k, D) for k = 0, 1, 2, ... (the oracle's MLP order)."""
    rng = np.random.Generator(np.random.PCG64(1))
    K = 64
    a = rng.standard_normal(K).astype(np.float32)
    b = (rng.standard_normal(K) * 10 ** rng.uniform(-3, 3, K)).astype(np.float32)
    c = np.float32(0.3)
    dev = native.math_selftest(100, np.concatenate([a, b, [c]]).astype(np.float64))[0]
    acc = np.float64(c)
    chain = np.float32(c)
    for k in range(K):
        # float32 fma: exact product in float64 (24+24 bits), one rounding after the add (double rounding is vanishingly rare)
        chain = np.float32(np.float64(a[k]) * np.float64(b[k]) + np.float64(chain))
    assert np.float32(dev) == chain, (dev, chain, acc)


def _kernel_form(e):
    """What the HIP engine's last search ran as (engine_host.h: 0 search kernel, 1 per-layer launches, 2 team kernel)."""
    return e.search_info()["kernel_form_id"]


_last_kernel_name = {}


def _run(engine_cls, kw, desc, blob, roots, carry=None, sidx=0, forms=None):
    e = engine_cls(**kw)
    e.set_weights(desc, blob)
    e.set_search_index(sidx)
    e.search(roots, carry)
    out = (e.results(), e.dump_tree(), e.root_children(), e.root_eval())
    if forms is not None:
        forms.append(_kernel_form(e))
        _last_kernel_name["name"] = e.search_info()["kernel_name"]
    e.close()
    return out


def _assert_same(a, b):
    for da, db in zip(a, b):
        if isinstance(da, dict):
            for k in da:
                np.testing.assert_array_equal(da[k], db[k], err_msg=k)
        else:
            for x, y in zip(da, db):
                np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("name", P.T1_NAMES)
def test_hip_matches_reference_goldens(native, name):
    case, z = P.load_case(name)
    out = P.run_case(native.HipEngine, case, z)
    P.compare_rows(out, z, float_tol=1e-12)
    # and bit for bit against the oracle on the same inputs
    ref = P.run_case(O.OracleEngine, case, z)
    for (r1, d1, c1), (r2, d2, c2) in zip(out, ref):
        for k in r1:
            np.testing.assert_array_equal(r1[k], r2[k], err_msg=k)
        for k in d1:
            np.testing.assert_array_equal(d1[k], d2[k], err_msg=k)
        np.testing.assert_array_equal(c1, c2)


CONFIGS = [
    # (env, mode, hidden, act, n_sims, extra)
    (2, 1, [256, 256], "elu", 200, dict(c_uct=0.05, gamma=1.0)),
    (2, 1, [256, 256], "elu", 64, dict(c_uct=0.3, gamma=0.97, c_pw=2.5, kappa=0.7, epsilon=0.2, v_target="on_policy")),
    (1, 1, [128, 128, 128], "elu", 90, dict(c_uct=0.1, gamma=0.99, c_pw=1.0, kappa=0.5)),
    (2, 1, [64], "relu", 50, dict(c_uct=0.05, gamma=1.0, c_pw=3.0, kappa=0.9)),
    (2, 1, [100, 60], "elu", 40, dict(c_uct=0.05, gamma=1.0)),
    (2, 1, [128, 128, 128, 128, 128], "relu", 30, dict(c_uct=0.05, gamma=1.0)),
    (0, 0, [128, 128], "relu", 100, dict(c_uct=1.5, gamma=1.0, num_actions=2)),
    (0, 0, [64, 64], "elu", 60, dict(c_uct=20.0, gamma=0.95, epsilon=0.1, num_actions=2, v_target="on_policy")),
    (0, 0, [256, 256], "relu", 80, dict(c_uct=5.0, gamma=0.99, num_actions=2)),
    # the other trunk nonlinearities
    (2, 1, [64, 64], "leakyrelu", 30, dict(c_uct=0.05, gamma=1.0)),
    (2, 1, [128, 128], "silu", 30, dict(c_uct=0.05, gamma=1.0)),
    (0, 0, [64, 64], "hardswish", 40, dict(c_uct=4.0, gamma=1.0, num_actions=2)),
    (0, 0, [64], "relu6", 40, dict(c_uct=4.0, gamma=1.0, num_actions=2)),
    # LayerNorm trunks (widths that are not multiples of 64 exercise the padded-unit mask)
    (2, 1, [100, 60], "elu", 30, dict(c_uct=0.05, gamma=1.0, _ln=True)),
    (2, 1, [256, 256], "relu", 30, dict(c_uct=0.05, gamma=1.0, _ln=True)),
    (0, 0, [128, 128, 128], "silu", 30, dict(c_uct=4.0, gamma=1.0, num_actions=2, _ln=True)),
    # Gaussian-mixture policy heads (the reference's default continuous config: 2 components, 3x128 ELU)
    (2, 1, [128, 128, 128], "elu", 60, dict(c_uct=0.05, gamma=1.0, _ncomp=2)),
    (1, 1, [64, 64], "elu", 40, dict(c_uct=0.2, gamma=0.95, c_pw=1.5, kappa=0.6, _ncomp=3)),
    # wide MLPs (BASELINE config E is 4x1024): weights streamed from L2, activation buffers up to 128 KB of LDS
    (2, 1, [512, 512], "elu", 30, dict(c_uct=0.05, gamma=1.0)),
    (2, 1, [512, 512, 512], "elu", 20, dict(c_uct=0.05, gamma=1.0, _ncomp=2)),
    (2, 1, [1024, 1024, 1024, 1024], "elu", 12, dict(c_uct=0.05, gamma=1.0)),
    (0, 0, [512], "relu", 40, dict(c_uct=3.0, gamma=1.0, num_actions=2)),
    # trees too large for LDS residency (> 255 records): global-memory tree storage
    (2, 1, [64, 64], "elu", 300, dict(c_uct=0.05, gamma=1.0)),
    (0, 0, [64, 64], "relu", 200, dict(c_uct=8.0, gamma=0.98, num_actions=2)),
    # up to 16 children per node: the LDS child-list pool grows through all its block sizes (4, 8, 16)
    (2, 1, [256, 256], "relu", 120, dict(c_uct=0.2, gamma=0.98, c_pw=1.4, kappa=0.5)),
    (1, 1, [64, 64], "elu", 254, dict(c_uct=0.02, gamma=1.0, c_pw=1.0, kappa=0.5)),
    # three actions (gym MountainCar-v0): the generic-A paths of evaluation, selection, re-scoring; LDS trees of 8- and 9-bit ids,
    # global trees, a 2x256 network (8-wave variants), a wide one (lock-step path), epsilon-greedy
    (3, 0, [64, 64], "relu", 60, dict(c_uct=0.8, gamma=0.99, num_actions=3)),
    (3, 0, [128, 128], "elu", 120, dict(c_uct=2.0, gamma=1.0, num_actions=3, epsilon=0.2, v_target="on_policy")),
    (3, 0, [256, 256], "relu", 50, dict(c_uct=1.5, gamma=0.97, num_actions=3)),
    (3, 0, [512, 512], "relu", 25, dict(c_uct=1.5, gamma=1.0, num_actions=3)),
    (3, 0, [64], "relu", 200, dict(c_uct=3.0, gamma=0.98, num_actions=3, v_target="greedy")),
    # MCTSContinuous over an env whose episodes END (gym MountainCarContinuous-v0; mcts.py:619-623, 682): terminal nodes in the continuous
    # descent -- 4-wave LDS kernels, the 8-wave shapes of 2x256 networks (lean walkers), a mixture head, a wide network (team kernel /
    # per-layer launches / one-launch kernel), trees in global memory (> 255 records), > 16 children per node
    (4, 1, [64, 64], "elu", 120, dict(c_uct=0.05, gamma=1.0, action_bound=1.0)),
    (4, 1, [256, 256], "elu", 150, dict(c_uct=0.1, gamma=0.98, epsilon=0.15, v_target="on_policy", action_bound=1.0)),
    (4, 1, [128, 128, 128], "elu", 60, dict(c_uct=0.05, gamma=1.0, action_bound=1.0, _ncomp=2)),
    (4, 1, [512, 512], "elu", 30, dict(c_uct=0.05, gamma=1.0, action_bound=1.0)),
    (4, 1, [64, 64], "relu", 300, dict(c_uct=0.05, gamma=0.99, action_bound=1.0)),
    (4, 1, [128, 128], "elu", 90, dict(c_uct=0.2, gamma=1.0, c_pw=2.0, kappa=0.6, action_bound=1.0, v_target="greedy")),
    # six observations (gym Acrobot-v1): two k-steps in the network's first layer, Runge-Kutta dynamics, reward 0 on the terminal step --
    # LDS trees of 8- and 9-bit ids, global trees, epsilon-greedy, 2x256 and a wide network (one-launch kernel: the team kernels take
    # at most four inputs), LayerNorm (weight-streaming kernels)
    (5, 0, [64, 64], "relu", 60, dict(c_uct=1.0, gamma=0.99, num_actions=3)),
    (5, 0, [128, 128], "elu", 120, dict(c_uct=2.0, gamma=1.0, num_actions=3, epsilon=0.2, v_target="on_policy")),
    (5, 0, [256, 256], "relu", 50, dict(c_uct=1.5, gamma=0.97, num_actions=3)),
    (5, 0, [512, 512], "relu", 20, dict(c_uct=1.5, gamma=1.0, num_actions=3)),
    (5, 0, [64], "relu", 200, dict(c_uct=3.0, gamma=0.98, num_actions=3, v_target="greedy")),
    (5, 0, [100, 60], "silu", 30, dict(c_uct=2.0, gamma=1.0, num_actions=3, _ln=True)),
]


def upswing_roots(roots):
    """Acrobot roots on the upswing (the synthetic ones hang at rest, hundreds of steps from the episode's end)."""
    return np.stack([1.4 + 6.0 * roots[:, 0], 8.0 * roots[:, 1], 3.0 + 20.0 * roots[:, 2], 20.0 * roots[:, 3]], 1)


def slope_roots(roots):
    """MountainCarContinuous roots on the slope below the flag (the synthetic ones rest in the valley, out of the flag's reach)."""
    u = (roots[:, 0] + 0.6) / 0.2
    return np.stack([0.25 + 0.199 * u, 0.02 + 0.05 * ((17.0 * u) % 1.0)], 1)


@pytest.mark.parametrize("cfg", CONFIGS, ids=[f"cfg{i}" for i in range(len(CONFIGS))])
@pytest.mark.parametrize("variant", ["default", "stream_weights", "global_tree", "persistent", "launches", "waves8", "waves4", "groups2",
                                     "trace_cap1", "trace_cap64", "tile16", "no_spec"])
def test_hip_bit_exact_vs_oracle(native, cfg, variant, monkeypatch):
    """Seeded batches (ragged: not a multiple of the 16-tree workgroup) -- every record of every tree must be identical.
    Variants force the other code paths: weights streamed from L2 instead of registers, trees in global memory
    instead of LDS, for wide networks (default: the persistent team kernel) the one-launch search kernel and the per-layer
    launches, and for 2x256 networks the workgroup shapes: eight waves / 16 trees (four of them walking: the default in continuous
    mode), four waves (the general shape) and eight waves / 32 trees (chosen by itself for batches of more 16-tree groups than CUs); in discrete mode one trace per simulation step (the round-3 loop)
    and as many as a tree can run without the network (default: at most five); full 16-tree tiles where small batches of small
    networks take half-filled ones."""
    env, mode, hidden, act, n_sims, extra = cfg
    extra = dict(extra)
    ncomp = extra.pop("_ncomp", 0)
    ln = extra.pop("_ln", False)
    if variant == "stream_weights":
        monkeypatch.setenv("AZG_FORCE_STREAM_WEIGHTS", "1")
    if variant == "global_tree":
        monkeypatch.setenv("AZG_FORCE_GLOBAL_TREE", "1")
    if variant in ("waves8", "waves4", "groups2"):
        if hidden != [256, 256] or ln or ncomp or mode != 1:
            pytest.skip("the 8-wave workgroups exist for 2x256 squashed-Normal networks (continuous mode)")
        monkeypatch.setenv(*(("AZG_WAVES", variant[-1]) if variant.startswith("waves") else ("AZG_GROUPS", "2")))
    if variant in ("trace_cap1", "trace_cap64"):
        if mode != 0 or max(hidden) > 256:
            pytest.skip("several traces per step: the discrete persistent search kernels")
        monkeypatch.setenv("AZG_TRACE_CAP", variant[len("trace_cap"):])
    if variant == "no_spec":
        # register-resident one-layer networks with common parameters run kernels specialised at compile time (dispatch.cuh: SPEC);
        # this variant forces the general kernels on the same inputs
        # (SPEC kernels keep their trees in LDS with 8-bit ids: at most 255 records per tree -- n_sims + 2 in continuous mode, the root
        # plus two edges per evaluated node in CartPole's discrete mode; cfg0, the headline shape with its 202 records, is one of them)
        records = n_sims + 2 if mode == 1 else 1 + 2 * (n_sims + 1)
        if len(hidden) != 2 or max(hidden) > 256 or ln or ncomp or extra.get("epsilon", 0.0) != 0.0 or records > 255 or env in (3, 5):
            pytest.skip("no compile-time specialised kernel exists for this configuration")
        monkeypatch.setenv("AZG_NO_SPEC", "1")
    if variant == "tile16":
        if max(hidden) > 128 or len(hidden) != 2 or ln or ncomp:
            pytest.skip("half-filled tiles exist for register-resident networks up to 128 wide")
        monkeypatch.setenv("AZG_TILE_TREES", "16")
    if variant in ("persistent", "launches"):
        if max(hidden) <= 256:
            pytest.skip("lock-step kernels only exist for hidden widths >= 512")
        if variant == "persistent":
            monkeypatch.setenv("AZG_FORCE_PERSISTENT", "1")   # wide networks: the one-launch kernel instead of the lock-step path
        else:
            monkeypatch.setenv("AZG_LS_TEAM", "0")            # the per-layer launches instead of the persistent team kernel
    B = 37
    kw = dict(env_id=env, mode=mode, n_trees=B, n_sims=n_sims, seed=1234, tree_id_base=77, **extra)
    in_dim, n_dist = (2 if env == 4 else 3, 3 * ncomp if ncomp else 2) if mode == 1 else {3: (2, 3), 5: (6, 3)}.get(env, (4, 2))
    desc = _capi.make_desc(in_dim, hidden, n_dist, act, num_components=ncomp, layernorm=ln)
    blob = O.make_weights(99, in_dim, hidden, n_dist, scale=2.0)
    if ln:
        blob = O.add_layernorm(blob, in_dim, hidden, n_dist, 7)
    o = O.OracleEngine(**kw)
    roots = o.synthetic_roots()
    o.close()
    if env == 0:
        roots[3] = [2.35, 1.5, 0.0, 0.0]      # terminates quickly
        roots[5] = [0.0, 0.0, 0.2, 1.0]
    if env == 3:
        roots[3] = [0.44, 0.04]               # the flag is two steps away
        roots[5] = [-1.195, -0.05]            # into the left wall
    if env == 5:
        roots = upswing_roots(roots)
        for i in range(B):
            if O.env_step(5, roots[i], 1)[2] and (-np.cos(roots[i][0]) - np.cos(roots[i][1] + roots[i][0])) > 1.0:
                roots[i] = [1.0, 0.0, 0.5, 0.0]           # (already above the line)
        roots[3] = [1.9, 0.2, 2.0, 1.0]           # every torque swings the tip over the line: terminal children, reward 0
        roots[5] = [1.373, -0.681, 2.48, 1.698]   # three steps below it
        roots[7] = [0.05, -0.03, 0.02, 0.01]      # hanging at rest
    if env == 4:
        roots = slope_roots(roots)
        roots[3] = [0.44, 0.03]               # every action reaches the flag: a search of traces that end in terminal nodes
        roots[5] = [-1.195, -0.05]            # into the left wall
        roots[7] = [-0.5, 0.0]                # the valley: no terminal node within reach
    carry = (np.arange(B) % 7).astype(np.int32) if mode == 0 else None
    forms = []
    a = _run(native.HipEngine, kw, desc, blob, roots, carry, sidx=3, forms=forms)
    b = _run(O.OracleEngine, kw, desc, blob, roots, carry, sidx=3)
    _assert_same(a, b)
    if variant in ("default", "no_spec") and forms == [0]:
        spec = _last_kernel_name.get("name", "").rstrip(">").endswith(", 1")
        if variant == "no_spec":
            assert not spec, _last_kernel_name
    if env == 4:   # the point of these configurations: terminal nodes, and traces that ended in an existing one (no new record)
        assert ((a[1]["node_flags"] & 2) != 0).any(1).sum() >= B // 3 and (a[1]["n_records"] < n_sims + 1).sum() >= B // 3


def test_deep_discrete_traces(native):
    """CartPole, 2500 simulations with an exploration constant that finds the balancing line: traces 80+ levels deep, far beyond the
    16 levels a tree's lanes hold at once, so the backup (and, in discrete mode, the re-scoring of the path's selections) continues
    through the generic parent-link walk (trees of this size live in global memory).  Every record bit-exact against the oracle;
    the depth is asserted."""
    NS, B = 2500, 9
    kw = dict(env_id=0, mode=0, n_trees=B, n_sims=NS, c_uct=5.0, gamma=1.0, num_actions=2, seed=8, tree_id_base=5)
    desc = _capi.make_desc(4, [64, 64], 2, "relu")
    blob = O.make_weights(12, 4, [64, 64], 2, scale=0.2)
    roots = np.zeros((B, 4))
    roots[:, 2] = np.linspace(-0.01, 0.01, B)
    a = _run(native.HipEngine, kw, desc, blob, roots, None, sidx=2)
    b = _run(O.OracleEngine, kw, desc, blob, roots, None, sidx=2)
    _assert_same(a, b)
    par, nrec = a[1]["parent"], a[1]["n_records"]
    deepest = 0
    for t in range(B):
        depth = np.zeros(int(nrec[t]), np.int32)
        for j in range(1, int(nrec[t])):
            depth[j] = depth[par[t][j]] + 1
        deepest = max(deepest, int(depth.max()))
    assert deepest > 60, deepest


@pytest.mark.parametrize("big", [False, True])
def test_carried_root_counts_beyond_the_sqrt_table(native, big):
    """A reused root searched again and again without moving on (legal in the reference: act() twice on one state) carries a
    visit count beyond the host-built sqrt(n + 1) table (4 n_sims + 4 entries): the kernel then computes the root's square root
    in place.  `big`: counts that no longer fit the 16-bit LDS records -> global-memory trees."""
    NS, B = 40, 19
    kw = dict(env_id=0, mode=0, n_trees=B, n_sims=NS, c_uct=30.0, gamma=0.98, num_actions=2, seed=5, tree_id_base=3)
    desc = _capi.make_desc(4, [64, 64], 2, "relu")
    blob = O.make_weights(9, 4, [64, 64], 2, scale=2.0)
    o = O.OracleEngine(**kw)
    roots = o.synthetic_roots()
    o.close()
    carry = (np.arange(B) * 37 % 400).astype(np.int32)       # 4 * NS + 4 = 164: both sides of the table's end
    if big:
        carry[::3] += 70000
    a = _run(native.HipEngine, kw, desc, blob, roots, carry, sidx=1)
    b = _run(O.OracleEngine, kw, desc, blob, roots, carry, sidx=1)
    _assert_same(a, b)
    np.testing.assert_array_equal(a[1]["node_n"][:, 0], carry + NS)


def test_terminal_root_raises(native):
    e = native.HipEngine(env_id=0, mode=0, n_trees=2, n_sims=4, c_uct=1.5, gamma=1.0, num_actions=2)
    e.set_weights(_capi.make_desc(4, [64], 2, "relu"), O.make_weights(1, 4, [64], 2))
    with pytest.raises(ValueError):
        e.search(np.array([[0.0, 0, 0, 0], [3.0, 0, 0, 0]]))
    e.close()
    # continuous mode (mcts.py:599-600): a MountainCarContinuous root at the flag
    e = native.HipEngine(env_id=4, mode=1, n_trees=2, n_sims=4, c_uct=0.05, gamma=1.0, action_bound=1.0)
    e.set_weights(_capi.make_desc(2, [64], 2, "relu"), O.make_weights(1, 2, [64], 2))
    with pytest.raises(ValueError):
        e.search(np.array([[-0.5, 0.0], [0.46, 0.01]]))
    e.search(np.array([[-0.5, 0.0], [0.46, -0.01]]))    # beyond the flag but rolling back: not terminal (gym: velocity >= 0)
    e.close()


def _assert_block_identical(r, d, ro, do, lo, hi):
    """Every result row and every tree record of trees [lo, hi): HIP engine (r, d: whole batch) against the oracle's block."""
    for k in ro:
        np.testing.assert_array_equal(ro[k], r[k][lo:hi], err_msg=f"{k} trees {lo}..{hi}")
    for k in do:
        np.testing.assert_array_equal(do[k], d[k][lo:hi], err_msg=f"{k} trees {lo}..{hi}")


@pytest.mark.parametrize("B,no_spec", [(4096, False), (4096, True), (8192 + 40, False)], ids=["4096", "4096-no_spec", "8232"])
def test_full_size_properties(native, B, no_spec, monkeypatch):
    """BASELINE config C (Pendulum-v1, 4096 trees, n_sims 200, 2x256 elu; mcts.py:656-702): size-independent invariants
     on every tree, and EVERY record of EVERY tree (results + whole tree dump) bit-exact against the oracle.
    The larger, ragged batch takes the 8-wave / 32-tree workgroups.  no_spec: the general kernel (AZG_NO_SPEC=1) instead of the
    compile-time specialised one the headline runs on -- the fallback path at the headline's size."""
    NS = 200
    if no_spec:
        monkeypatch.setenv("AZG_NO_SPEC", "1")
    kw = dict(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    desc = _capi.make_desc(3, [256, 256], 2, "elu")
    blob = O.make_weights(34, 3, [256, 256], 2)
    e = native.HipEngine(**kw)
    e.set_weights(desc, blob)
    roots = e.synthetic_roots()
    e.search(roots)
    r, d = e.results(), e.dump_tree()
    assert e.search_info()["spec"] == (0 if no_spec else 1)
    e.close()
    assert (r["counts"].sum(1) == NS).all()                       # sum of root counts == n_sims
    assert (r["n_children"] == 15).all()                          # ceil(sqrt(200)) children at the root
    assert (d["n_records"] == NS + 1).all()                       # one new node per trace
    assert (d["node_n"][:, 0] == NS).all()
    # node.n == sum of child edge counts, for every node of every tree
    acc = np.zeros_like(d["node_n"])
    rows = np.repeat(np.arange(B), NS)
    np.add.at(acc, (rows, d["parent"][:, 1:NS + 1].ravel()), d["edge_n"][:, 1:NS + 1].ravel())
    np.testing.assert_array_equal(acc[:, :NS + 1], d["node_n"][:, :NS + 1])
    # the whole batch on the oracle (same global tree ids -> same noise), in blocks so that its trees stay cache-sized
    for lo in range(0, B, 1024):
        hi = min(B, lo + 1024)
        ro, do = _oracle_block(kw, desc, blob, roots, lo, hi)
        _assert_block_identical(r, d, ro, do, lo, hi)


def _tree_invariants(r, d, NS, trees):
    assert (r["counts"].sum(1) == NS).all()                       # sum of root counts == n_sims
    assert (d["n_records"] == NS + 1).all()                       # one new node per trace (Pendulum never terminates)
    assert (d["node_n"][:, 0] == NS).all()
    for t in trees:                                               # node.n == sum of child edge counts, for every node
        par, en, nn = d["parent"][t], d["edge_n"][t], d["node_n"][t]
        acc = np.zeros_like(nn)
        np.add.at(acc, par[1:NS + 1], en[1:NS + 1])
        np.testing.assert_array_equal(acc[:NS + 1], nn[:NS + 1])


def _oracle_block(kw, desc, blob, roots, lo, hi):
    """Trees [lo, hi) of a batch on the oracle (same global tree ids -> same noise), OpenMP over the block's trees."""
    oo = O.OracleEngine(**dict(kw, n_trees=hi - lo, tree_id_base=kw.get("tree_id_base", 0) + lo))
    oo.set_weights(desc, blob)
    oo.search(roots[lo:hi])
    out = oo.results(), oo.dump_tree()
    oo.close()
    return out


def test_config_b_full_size(native):
    """BASELINE config B (CartPole-v1 discrete, 4096 trees, n_sims 100, 2x128 relu) at size: count invariants on every tree (terminal
    leaves included: a trace that ends in a terminal node creates no record) and every record of all 4096 trees bit-exact
    against the oracle (mcts.py:418-462)."""
    NS, B = 100, 4096
    kw = dict(env_id=0, mode=0, n_trees=B, n_sims=NS, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)
    desc = _capi.make_desc(4, [128, 128], 2, "relu")
    blob = O.make_weights(34, 4, [128, 128], 2)
    e = native.HipEngine(**kw)
    e.set_weights(desc, blob)
    roots = e.synthetic_roots()
    e.search(roots)
    r, d = e.results(), e.dump_tree()
    assert _kernel_form(e) == 0
    e.close()
    assert (r["counts"].sum(1) == NS).all() and (r["n_children"] == 2).all()
    assert (d["node_n"][:, 0] == NS).all()
    assert (d["n_records"] <= 1 + 2 * (NS + 1)).all() and (d["n_records"] % 2 == 1).all()   # the root + two edges per expanded node
    for t in range(B):                                            # node.n == sum of child edge counts unless the node is terminal
        n = int(d["n_records"][t])
        par, en, nn, fl = d["parent"][t][:n], d["edge_n"][t][:n], d["node_n"][t][:n], d["node_flags"][t][:n]
        acc = np.zeros_like(nn)
        np.add.at(acc, par[1:], en[1:])
        inner = (fl & 2) == 0                                     # (FLAG_TERMINAL = 2: visits of a terminal node stop there)
        np.testing.assert_array_equal(acc[inner], nn[inner])
    for lo in range(0, B, 1024):                                  # every record of every tree against the oracle
        ro, do = _oracle_block(kw, desc, blob, roots, lo, lo + 1024)
        _assert_block_identical(r, d, ro, do, lo, lo + 1024)


def test_config_e_full_size_lockstep(native):
    """BASELINE config E per GPU (Pendulum-v1, 1024 trees, n_sims 200, 4x1024 ELU: mcts.py:656-702 at E's tree sizes) on the
    lock-step path (by default the persistent team kernel: 201 simulation steps in one launch).  Size-independent invariants on
    every tree + every record of all 1024 trees bit-exact against the oracle (results and whole tree dumps)."""
    NS, B = 200, 1024
    kw = dict(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    desc = _capi.make_desc(3, [1024] * 4, 2, "elu")
    blob = O.make_weights(34, 3, [1024] * 4, 2)
    e = native.HipEngine(**kw)
    e.set_weights(desc, blob)
    roots = e.synthetic_roots()
    e.search(roots)
    r, d = e.results(), e.dump_tree()
    e.close()
    assert (r["n_children"] == 15).all()
    _tree_invariants(r, d, NS, range(B))
    depth = np.zeros((B, NS + 1), np.int32)
    for j in range(1, NS + 1):
        depth[:, j] = depth[np.arange(B), d["parent"][:, j]] + 1
    assert depth.max() >= 4, depth.max()                          # real trees: several levels below the root
    for lo in range(0, B, 256):                                   # all 1024 trees (the oracle streams 12.6 MB of weights per evaluation:
        ro, do = _oracle_block(kw, desc, blob, roots, lo, lo + 256)   # about a minute on 16 host threads)
        _assert_block_identical(r, d, ro, do, lo, lo + 256)


def test_team_kernel_gives_up_instead_of_hanging(native, monkeypatch):
    """The persistent team kernel's waits are bounded (its workgroups must all be resident: another process on the GPU can
    prevent that).  With a spin limit of zero every wait counts as timed out: the launch leaves, the engine notices, redoes the
    search with the per-layer launches under the same search index, and stays on them -- the caller sees complete, identical
    results.  Also inside a self-play step, whose final-action kernel must not consume an abandoned search."""
    import ctypes as C
    kw = dict(env_id=2, mode=1, n_trees=64, n_sims=20, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=3)
    desc = _capi.make_desc(3, [512, 512], 2, "elu")
    blob = O.make_weights(5, 3, [512, 512], 2)

    def run(selfplay):
        e = native.HipEngine(**kw)
        e.set_weights(desc, blob)
        if selfplay:
            e.selfplay_begin(50, capacity_steps=3)
            for _ in range(3):
                e.selfplay_step()
            out = {"rows": e.selfplay_rows(clear=False)}
        else:
            e.search(e.synthetic_roots())
            e.search(e.synthetic_roots())
            out = dict(e.results(), **e.dump_tree())
        n = e.search_info()["team_fallbacks"]
        e.close()
        return out, n

    for selfplay in (False, True):
        monkeypatch.delenv("AZG_TEAM_SPIN_LIMIT", raising=False)
        want, n0 = run(selfplay)
        monkeypatch.setenv("AZG_TEAM_SPIN_LIMIT", "0")
        got, n1 = run(selfplay)
        assert n0 == 0 and n1 == 1, (n0, n1)      # the first search fell back, the later ones went straight to the launches
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_team_kernel_in_two_launches_gives_up_as_one(native, monkeypatch):
    """A batch that runs as two team launches (4136 trees, 4x1024): with a spin limit of zero the first launch raises the abort flag, the
    second one leaves at its first wait, and the engine redoes the whole search with the per-layer launches: identical results."""
    import ctypes as C
    kw = dict(env_id=2, mode=1, n_trees=4136, n_sims=6, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
    desc = _capi.make_desc(3, [1024] * 4, 2, "elu")
    blob = O.make_weights(34, 3, [1024] * 4, 2)

    def run():
        e = native.HipEngine(**kw)
        e.set_weights(desc, blob)
        e.search(e.synthetic_roots())
        form = _kernel_form(e)
        out = dict(e.results(), **e.dump_tree())
        n = e.search_info()["team_fallbacks"]
        e.close()
        return out, n, form

    monkeypatch.delenv("AZG_TEAM_SPIN_LIMIT", raising=False)
    want, n0, f0 = run()
    monkeypatch.setenv("AZG_TEAM_SPIN_LIMIT", "0")
    got, n1, f1 = run()
    assert (n0, f0) == (0, 2) and (n1, f1) == (1, 1), (n0, f0, n1, f1)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_config_e_persistent_kernel_equals_lockstep(native, monkeypatch):
    """The same network and search on the one-launch kernel (AZG_FORCE_PERSISTENT=1; weights streamed from L2) on a smaller batch:
    identical trees to the lock-step path's, record for record."""
    NS, B = 200, 80
    kw = dict(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34, tree_id_base=512)
    desc = _capi.make_desc(3, [1024] * 4, 2, "elu")
    blob = O.make_weights(34, 3, [1024] * 4, 2)
    e = native.HipEngine(**kw)
    roots = e.synthetic_roots()
    e.close()
    a = _run(native.HipEngine, kw, desc, blob, roots)
    monkeypatch.setenv("AZG_FORCE_PERSISTENT", "1")
    b = _run(native.HipEngine, kw, desc, blob, roots)
    _assert_same(a, b)
    _tree_invariants(a[0], a[1], NS, range(B))
    ro, do = _oracle_block(kw, desc, blob, roots, 70, 74)
    for k in do:
        np.testing.assert_array_equal(do[k], b[1][k][70:74], err_msg=k)


@pytest.mark.parametrize("cfg", ["C", "B"])
def test_shard_invariance_on_the_device(native, cfg):
    """SURVEY 4.4 / 8e on the HIP engine: a tree's result depends on its GLOBAL id only.  One engine of 4096 trees (config D's
    per-GPU leg) == two engines of 2048 trees with tree_id_base 0 and 2048, row for row and record for record."""
    if cfg == "C":
        kw = dict(env_id=2, mode=1, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)
        desc, blob = _capi.make_desc(3, [256, 256], 2, "elu"), O.make_weights(34, 3, [256, 256], 2)
    else:
        kw = dict(env_id=0, mode=0, n_sims=100, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)
        desc, blob = _capi.make_desc(4, [128, 128], 2, "relu"), O.make_weights(34, 4, [128, 128], 2)
    whole = native.HipEngine(n_trees=4096, **kw)
    roots = whole.synthetic_roots()
    whole.close()
    full = _run(native.HipEngine, dict(kw, n_trees=4096, tree_id_base=0), desc, blob, roots)
    for lo in (0, 2048):
        e = native.HipEngine(n_trees=2048, tree_id_base=lo, **kw)
        np.testing.assert_array_equal(e.synthetic_roots(), roots[lo:lo + 2048])   # synthetic roots are keyed by the global id too
        e.close()
        part = _run(native.HipEngine, dict(kw, n_trees=2048, tree_id_base=lo), desc, blob, roots[lo:lo + 2048])
        for df, dp in zip(full, part):
            if isinstance(df, dict):
                for k in df:
                    np.testing.assert_array_equal(df[k][lo:lo + 2048], dp[k], err_msg=k)
            else:
                for x, y in zip(df, dp):
                    np.testing.assert_array_equal(x[lo:lo + 2048], y)


@pytest.mark.parametrize("cfg", ["D", "E"])
def test_last_rank_of_an_8_gpu_job(native, cfg):
    """The N = 8 configurations that no single box can run whole (BASELINE configs D: 32768 games, and E: 8192 trees of the 4x1024
    network, sharded over 8 GPUs): the leg of the LAST rank -- global tree ids 7 * B/8 ... B - 1 through `tree_id_base`, exactly as
    bench.py and distributed.shard_range set it up -- on this one GPU, against the oracle given the same global ids.  Synthetic
    roots, noise streams and results are keyed by the global id, so this is the same computation rank 7 of the real job runs."""
    from alphazero_gym_amd import distributed as D
    if cfg == "D":
        total, hidden = 32768, [256, 256]
    else:
        total, hidden = 8192, [1024] * 4
    lo, hi = D.shard_range(total, 7, 8)
    assert (lo, hi) == (7 * total // 8, total)
    kw = dict(env_id=2, mode=1, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34, n_trees=hi - lo, tree_id_base=lo)
    desc, blob = _capi.make_desc(3, hidden, 2, "elu"), O.make_weights(34, 3, hidden, 2)
    e = native.HipEngine(**kw)
    roots = e.synthetic_roots()
    e.set_weights(desc, blob)
    e.search(roots)
    r, d = e.results(), e.dump_tree()
    e.close()
    assert (r["counts"].sum(1) == 200).all()
    n = 512 if cfg == "D" else 128                                  # the shard's last trees against the oracle (all of them take minutes on the host)
    okw = dict(kw, n_trees=n, tree_id_base=hi - n)
    o = O.OracleEngine(**okw)
    np.testing.assert_array_equal(o.synthetic_roots(), roots[-n:])
    o.set_weights(desc, blob)
    o.search(roots[-n:])
    ro, do = o.results(), o.dump_tree()
    o.close()
    for k in ro:
        np.testing.assert_array_equal(ro[k], r[k][-n:], err_msg=k)
    for k in do:
        np.testing.assert_array_equal(do[k], d[k][-n:], err_msg=k)


def _random_case(rng):
    """A random engine configuration + network: every mode, width class, activation, head and search parameter."""
    cont = rng.random() < 0.5
    n_hidden = int(rng.integers(1, 5))
    width = int(rng.choice([24, 64, 100, 128, 200, 256]))
    hidden = [width] * n_hidden if rng.random() < 0.7 else [int(rng.choice([48, 64, 128, 256])) for _ in range(n_hidden)]
    act = str(rng.choice(["relu", "elu", "elu", "relu", "silu", "leakyrelu", "relu6", "hardswish"]))
    ln = bool(rng.random() < 0.15)
    n_sims = int(rng.choice([1, 2, 7, 33, 64, 120, 253, 254, 300]))
    extra = dict(gamma=float(rng.choice([1.0, 0.99, 0.9])), epsilon=float(rng.choice([0.0, 0.0, 0.25])),
                 v_target=str(rng.choice(["off_policy", "on_policy", "greedy"])))
    ncomp = 0
    if cont:
        env = int(rng.choice([1, 2, 4]))   # (4: MountainCarContinuous -- terminal nodes in the continuous search)
        extra.update(c_uct=float(rng.choice([0.02, 0.05, 0.5])), c_pw=float(rng.choice([0.5, 1.0, 1.13, 2.0, 4.0])),
                     kappa=float(rng.choice([0.3, 0.5, 0.75])))
        if env == 4:
            extra["action_bound"] = float(rng.choice([1.0, 1.0, 2.0]))
        if rng.random() < 0.25:
            ncomp = int(rng.integers(2, 6))
        in_dim, n_dist = (2 if env == 4 else 3), (3 * ncomp if ncomp else 2)
        mode = 1
    elif rng.random() < 0.65:
        env, mode, in_dim, n_dist = 0, 0, 4, 2
        extra.update(c_uct=float(rng.choice([1.5, 5.0, 30.0])), num_actions=2)
    elif rng.random() < 0.5:   # three actions (gym MountainCar-v0)
        env, mode, in_dim, n_dist = 3, 0, 2, 3
        extra.update(c_uct=float(rng.choice([0.8, 2.0, 6.0])), num_actions=3)
    else:   # three actions, six observations (gym Acrobot-v1)
        env, mode, in_dim, n_dist = 5, 0, 6, 3
        extra.update(c_uct=float(rng.choice([0.8, 2.0, 6.0])), num_actions=3)
    if mode == 0 and rng.random() < 0.2:
        extra["tie_break"] = "random"
    return env, mode, hidden, act, ln, n_sims, extra, ncomp, in_dim, n_dist


@pytest.mark.parametrize("seed", range(32))
def test_hip_bit_exact_vs_oracle_random_configurations(native, seed):
    """Seeded random configurations (modes, widths, depths, activations, heads, widening laws, tree sizes on both sides of the
    LDS limits, eps-greedy, value targets): every record of every tree identical to the oracle's."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    env, mode, hidden, act, ln, n_sims, extra, ncomp, in_dim, n_dist = _random_case(rng)
    B = int(rng.choice([1, 5, 16, 19, 33]))
    if os.environ.get("AZG_FUZZ_BIG"):      # tests/fuzz_parity.py: batches of several workgroups (ragged), 32-tree workgroups
        B = int(rng.choice([19, 70, 130, 257, 515]))
    kw = dict(env_id=env, mode=mode, n_trees=B, n_sims=n_sims, seed=int(rng.integers(1, 1 << 30)), tree_id_base=int(rng.integers(0, 1000)), **extra)
    if os.environ.get("AZG_FUZZ_SHAPES"):   # tests/fuzz_parity.py: also the 8-wave workgroup shapes, global trees, weights from L2
        os.environ.pop("AZG_WAVES", None); os.environ.pop("AZG_GROUPS", None); os.environ.pop("AZG_FORCE_GLOBAL_TREE", None); os.environ.pop("AZG_FORCE_STREAM_WEIGHTS", None)
        os.environ.pop("AZG_TRACE_CAP", None); os.environ.pop("AZG_TILE_TREES", None)
        cap = int(rng.choice([0, 0, 1, 2, 3, 7, 1000]))
        if cap: os.environ["AZG_TRACE_CAP"] = str(cap)
        tile = int(rng.choice([0, 8, 16]))
        if tile: os.environ["AZG_TILE_TREES"] = str(tile)
        pick = int(rng.integers(0, 6))
        if pick == 1: os.environ["AZG_WAVES"] = "8"
        if pick == 2: os.environ["AZG_GROUPS"] = "2"
        if pick == 3: os.environ["AZG_FORCE_GLOBAL_TREE"] = "1"
        if pick == 4: os.environ["AZG_FORCE_STREAM_WEIGHTS"] = "1"
    desc = _capi.make_desc(in_dim, hidden, n_dist, act, num_components=ncomp, layernorm=ln)
    blob = O.make_weights(int(rng.integers(1, 1000)), in_dim, hidden, n_dist, scale=float(rng.choice([1.0, 2.0, 3.0])))
    if ln:
        blob = O.add_layernorm(blob, in_dim, hidden, n_dist, 7)
    o = O.OracleEngine(**kw)
    roots = o.synthetic_roots()
    o.close()
    if env == 4:
        roots = slope_roots(roots)
    if env == 5:
        roots = upswing_roots(roots)
        for i in range(B):
            if (-np.cos(roots[i][0]) - np.cos(roots[i][1] + roots[i][0])) > 0.98:
                roots[i] = [1.0, 0.0, 0.5, 0.0]
    carry = np.minimum(np.arange(B) % 5, 3 * n_sims).astype(np.int32) if mode == 0 else None
    sidx = int(rng.integers(0, 50))
    a = _run(native.HipEngine, kw, desc, blob, roots, carry, sidx=sidx)
    b = _run(O.OracleEngine, kw, desc, blob, roots, carry, sidx=sidx)
    _assert_same(a, b)


class _DevArr:
    """A raw device pointer as a CUDA-array-interface object (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


@pytest.mark.parametrize("hidden,act,ln,ncomp", [([256, 256], "elu", False, 0), ([1024] * 4, "elu", False, 0), ([100, 60], "relu", True, 0),
                                                 ([128, 128, 128], "elu", False, 2)])
def test_device_side_weight_sync_equals_the_host_path(native, hidden, act, ln, ncomp):
    """azg_set_weights_device (blob in HBM, re-layout by the engine's gather kernel) against azg_set_weights (host blob): the same
    network -- identical azg_mlp_eval outputs and identical trees -- also after a second, different blob through the cached map."""
    import torch
    n_dist = 3 * ncomp if ncomp else 2
    desc = _capi.make_desc(3, hidden, n_dist, act, num_components=ncomp, layernorm=ln)
    kw = dict(env_id=2, mode=1, n_trees=40, n_sims=12, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=9)
    host, dev = native.HipEngine(**kw), native.HipEngine(**kw)
    roots = host.synthetic_roots()
    obs = np.random.Generator(np.random.PCG64(3)).uniform(-1, 1, (50, 3)).astype(np.float32)
    for wseed in (5, 6):
        blob = O.make_weights(wseed, 3, hidden, n_dist, scale=2.0)
        if ln:
            blob = O.add_layernorm(blob, 3, hidden, n_dist, 7)
        host.set_weights(desc, blob)
        d_blob = torch.from_numpy(blob).cuda()
        torch.cuda.synchronize()
        dev.set_weights_device(desc, d_blob.data_ptr(), d_blob.numel())
        d_blob.zero_()                                                # the engine keeps its own copy
        for a, b in zip(host.mlp_eval(obs), dev.mlp_eval(obs)):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        host.search(roots); dev.search(roots)
        _assert_same((host.results(), host.dump_tree(), host.root_eval()), (dev.results(), dev.dump_tree(), dev.root_eval()))
    host.close(); dev.close()


def test_set_policy_with_parameters_on_the_gpu_takes_the_device_path(native, monkeypatch):
    """Engine.set_policy with a torch policy living on the engine's GPU flattens it there and calls azg_set_weights_device; the
    result equals the host path's (policy_blob + azg_set_weights)."""
    import torch
    from alphazero_gym_amd.network.policies import make_policy
    torch.manual_seed(4)
    pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=[256, 256], nonlinearity="elu",
                      num_components=1, action_bound=2.0)
    kw = dict(env_id=2, mode=1, n_trees=20, n_sims=10, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=9)
    host, dev = native.HipEngine(**kw), native.HipEngine(**kw)
    host.set_policy(pol)                                              # CPU parameters: host path
    calls = []
    orig = dev.set_weights_device
    monkeypatch.setattr(dev, "set_weights_device", lambda *a: (calls.append(a[2]), orig(*a))[1])
    dev.set_policy(pol.cuda())
    assert calls == [sum(p.numel() for p in pol.parameters())]
    roots = host.synthetic_roots()
    host.search(roots); dev.search(roots)
    _assert_same((host.results(), host.dump_tree(), host.root_eval()), (dev.results(), dev.dump_tree(), dev.root_eval()))
    host.close(); dev.close()


def test_results_resident_hands_out_the_same_numbers_without_a_copy(native):
    """azg_results_resident: return_results (mcts.py:269-307) into the engine's device buffers, wrapped as torch tensors through
    the CUDA array interface: same values as the host download of azg_results."""
    import torch
    kw = dict(env_id=2, mode=1, n_trees=300, n_sims=40, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=9)
    e = native.HipEngine(**kw)
    e.set_weights(_capi.make_desc(3, [64, 64], 2, "elu"), O.make_weights(5, 3, [64, 64], 2))
    e.upload_roots(e.synthetic_roots())
    e.search_resident()
    p = e.results_resident()                                          # launches only
    e.sync()
    B, K = e.n_trees, e.kmax
    got = {"actions": torch.as_tensor(_DevArr(p["actions"], (B, K), "<f4"), device="cuda"),
           "counts": torch.as_tensor(_DevArr(p["counts"], (B, K), "<i4"), device="cuda"),
           "Q": torch.as_tensor(_DevArr(p["Q"], (B, K), "<f8"), device="cuda"),
           "v_target": torch.as_tensor(_DevArr(p["v_target"], (B,), "<f8"), device="cuda"),
           "n_children": torch.as_tensor(_DevArr(p["n_children"], (B,), "<i4"), device="cuda")}
    want = e.results()
    for k in want:
        np.testing.assert_array_equal(got[k].cpu().numpy(), want[k], err_msg=k)
    e.close()


TIE_CASES = [
    dict(env_id=0, mode=0, n_sims=60, c_uct=1.5, gamma=1.0, num_actions=2),
    dict(env_id=3, mode=0, n_sims=80, c_uct=1.0, gamma=1.0, num_actions=3),
    dict(env_id=3, mode=0, n_sims=50, c_uct=1.0, gamma=0.97, num_actions=3, epsilon=0.2),
    dict(env_id=2, mode=1, n_sims=60, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5),
]


def _tie_engine(cls, kw, tie, B=21):
    """A network of zeros: every value is 0 and every prior uniform, so exactly equal selection scores are the rule."""
    e = cls(n_trees=B, seed=77, tree_id_base=11, tie_break=tie, **kw)
    cont = kw["mode"] == 1
    in_dim, n_dist = (3, 2) if cont else ((2, 3) if kw["env_id"] == 3 else (4, 2))
    e.set_weights(_capi.make_desc(in_dim, [64, 64], n_dist, "relu"), np.zeros_like(O.make_weights(1, in_dim, [64, 64], n_dist)))
    roots = e.synthetic_roots()
    e.search(roots)
    out = (e.results(), e.dump_tree())
    e.close()
    return out


@pytest.mark.parametrize("kw", TIE_CASES, ids=["cartpole", "mountaincar", "mountaincar_eps", "pendulum"])
def test_random_tie_break_is_the_same_on_the_device_and_the_oracle(native, kw):
    """AZG_TIE_RANDOM (helpers.py:46-52's random.choice among exactly equal scores, as a Philox-keyed draw): identical trees on
    the HIP engine and on the oracle, and different from the lowest-index rule wherever ties occur."""
    a = _tie_engine(native.HipEngine, kw, "random")
    b = _tie_engine(O.OracleEngine, kw, "random")
    _assert_same(a, b)
    first = _tie_engine(native.HipEngine, kw, "first")
    _assert_same(first, _tie_engine(O.OracleEngine, kw, "first"))
    if kw["mode"] == 0:
        assert not np.array_equal(a[0]["counts"], first[0]["counts"])   # (Pendulum's sampled actions make ties rare: no claim there)


@pytest.mark.parametrize("B,eps", [(1536, 0.0), (2048, 0.0), (2000, 0.1), (3072, 0.0), (4136, 0.0)])
def test_team_kernel_beyond_two_workgroups_per_cu(native, B, eps, monkeypatch):
    """Batches beyond two 32-tree team workgroups per CU (config E's network with more than 1024 trees per GPU) run the team kernel's
    other forms: three short-chunk workgroups per CU (1536 trees), then teams of 64 trees -- 64 x 64 tiles, a workgroup walks four
    trees -- two (2048; specialised and, with eps-greedy, general tree phases) and three (3072) per CU; beyond that the batch runs as
    several launches of equal parts (4136 trees: two).  Same arithmetic as the
    per-layer launches: identical trees, record for record (the launches are checked against the oracle at config E's size above); a
    ragged slice of the batch also against the oracle."""
    NS = 24
    kw = dict(env_id=2, mode=1, n_trees=B, n_sims=NS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34, epsilon=eps)
    desc = _capi.make_desc(3, [1024] * 4, 2, "elu")
    blob = O.make_weights(34, 3, [1024] * 4, 2)
    e = native.HipEngine(**kw)
    roots = e.synthetic_roots()
    e.close()
    forms = []
    a = _run(native.HipEngine, kw, desc, blob, roots, forms=forms)
    monkeypatch.setenv("AZG_TEAM_WIDE", "0")
    b = _run(native.HipEngine, kw, desc, blob, roots, forms=forms)
    assert forms == [2, 1], forms                                  # team kernel, then the per-layer launches
    _assert_same(a, b)
    ro, do = _oracle_block(kw, desc, blob, roots, B - 40, B)
    _assert_block_identical(a[0], a[1], ro, do, B - 40, B)
