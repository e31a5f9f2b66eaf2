"""Device-resident self-play (azg_selfplay_*): the reference's run loops as golden vectors (tier T7) against the oracle (CPU)
and the HIP engine (GPU), the replay ring's FIFO rule against the reference's ReplayBuffer (T6), invariants, and bit-exact
HIP-vs-oracle rows."""
import ast
import glob
import os

import numpy as np
import pytest

import oracle_lib as O
from alphazero_gym_amd import _capi

CASES = {
    "pendulum": dict(kw=dict(env_id=2, mode=1, n_sims=30, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=11, tree_id_base=5),
                     desc=(3, [64, 64], 2, "elu"), max_len=4, det=False),
    "cartpole_sampled": dict(kw=dict(env_id=0, mode=0, n_sims=24, c_uct=20.0, gamma=0.97, num_actions=2, seed=12),
                             desc=(4, [64, 64], 2, "relu"), max_len=6, det=False),
    "cartpole_det": dict(kw=dict(env_id=0, mode=0, n_sims=24, c_uct=20.0, gamma=0.97, num_actions=2, seed=13, v_target="on_policy"),
                         desc=(4, [128, 128], 2, "relu"), max_len=7, det=True),
    # MountainCarContinuous: games start below the flag (roots uploaded after selfplay_begin), so episodes end by reaching it
    "mcc": dict(kw=dict(env_id=4, mode=1, n_sims=30, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=14, tree_id_base=9, action_bound=1.0),
                desc=(2, [64, 64], 2, "elu"), max_len=5, det=False, slope=True),
}


def slope_roots(e):
    """MountainCar roots on the slope below the flag instead of the valley (the same map as the T3-scale leg's)."""
    u = (e.synthetic_roots()[:, 0] + 0.6) / 0.2
    return np.stack([0.25 + 0.199 * u, 0.02 + 0.05 * ((17.0 * u) % 1.0)], 1)


def play(engine_cls, case, n_trees=21, steps=9):
    c = CASES[case]
    e = engine_cls(n_trees=n_trees, **c["kw"])
    in_dim, hidden, nd, act = c["desc"]
    e.set_weights(_capi.make_desc(in_dim, hidden, nd, act), O.make_weights(3, in_dim, hidden, nd, scale=2.0))
    e.selfplay_begin(c["max_len"], c["det"], capacity_steps=steps)
    if c.get("slope"):
        e.upload_roots(slope_roots(e))
    for _ in range(steps):
        e.selfplay_step()
    rows = e.selfplay_rows(clear=True)
    stats = e.selfplay_stats()
    assert e.selfplay_rows().shape[0] == 0   # cleared
    e.close()
    return rows, stats


@pytest.mark.parametrize("case", sorted(CASES))
def test_selfplay_invariants_on_oracle(case):
    rows, (fsum, fcnt, state) = play(O.OracleEngine, case)
    c = CASES[case]
    n_sims = c["kw"]["n_sims"]
    K = 6 if c["kw"]["mode"] == 1 else 2    # ceil(sqrt(30)) root children / two CartPole actions
    so = (2 if case == "mcc" else 3) if c["kw"]["mode"] == 1 else 4
    assert rows.shape == (9 * 21, so + 3 * K + 1)
    counts = rows[:, so + K:so + 2 * K]
    np.testing.assert_array_equal(counts.sum(1), np.full(len(rows), float(n_sims)))
    if case == "mcc":
        reached = fsum > 50                                                  # +100 at the flag
        assert reached.any() and (fcnt >= 1).all() and (np.abs(rows[:, so:so + K]) <= 1.0).all()   # actions within the bound
        assert (-0.6 <= state[reached, 0]).all() and (state[:, 0] < 0.45).all()   # a game that reached the flag restarted in the valley
    elif c["kw"]["mode"] == 1:
        np.testing.assert_allclose(np.hypot(rows[:, 0], rows[:, 1]), 1.0, atol=1e-6)
        assert (fcnt == 9 // c["max_len"]).all() and (fsum < 0).all()       # Pendulum never terminates: episodes end by length
    else:
        assert fcnt.sum() > 0 and (fsum[fcnt > 0] / fcnt[fcnt > 0] >= 1).all()
    assert np.isfinite(state).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_selfplay_hip_matches_oracle_bit_for_bit(case):
    from alphazero_gym_amd import _native
    a_rows, a_stats = play(_native.HipEngine, case)
    b_rows, b_stats = play(O.OracleEngine, case)
    np.testing.assert_array_equal(a_rows.view(np.uint32), b_rows.view(np.uint32))
    for x, y in zip(a_stats, b_stats):
        np.testing.assert_array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["config_D_leg", "cartpole_4096"])
def test_selfplay_at_baseline_size_hip_matches_oracle(which):
    """The device-resident self-play step at BASELINE size -- config D's per-GPU leg (Pendulum-v1, 4096 games, 200 simulations per
    move, 2x256 ELU; tree ids of rank 3 of 8) and CartPole with config B's search (4096 games, 100 simulations, 2x128 ReLU; several
    traces per step, episodes ending and restarting inside the window): every replay row of every game and step, the episode
    statistics and the games' env states identical to the oracle's, bit for bit."""
    from alphazero_gym_amd import _native
    if which == "config_D_leg":
        kw = dict(env_id=2, mode=1, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34, tree_id_base=3 * 4096)
        desc, steps, max_len = (3, [256, 256], 2, "elu"), 3, 2
    else:
        kw = dict(env_id=0, mode=0, n_sims=100, c_uct=1.5, gamma=1.0, num_actions=2, seed=34)
        desc, steps, max_len = (4, [128, 128], 2, "relu"), 6, 4
    out = []
    for cls in (_native.HipEngine, O.OracleEngine):
        e = cls(n_trees=4096, **kw)
        in_dim, hidden, nd, act = desc
        e.set_weights(_capi.make_desc(in_dim, hidden, nd, act), O.make_weights(34, in_dim, hidden, nd))
        e.selfplay_begin(max_len, False, capacity_steps=steps)
        for _ in range(steps):
            e.selfplay_step()
        out.append((e.selfplay_rows(clear=True), e.selfplay_stats()))
        e.close()
    (a_rows, a_stats), (b_rows, b_stats) = out
    assert a_rows.shape[0] == steps * 4096
    np.testing.assert_array_equal(a_rows.view(np.uint32), b_rows.view(np.uint32))
    for x, y in zip(a_stats, b_stats):
        np.testing.assert_array_equal(x, y)
    if which == "cartpole_4096":
        assert a_stats[1].sum() > 0      # episodes did end (and restart) inside the window


# ------------------------------------------------------------------------------------------------ T7: the reference's run loops

T7 = sorted(os.path.basename(p)[len("t7_selfplay_"):-4] for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t7_selfplay_*.npz")))
ENGINES = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


def _engine_cls(which):
    if which == "oracle":
        return O.OracleEngine
    from alphazero_gym_amd import _native
    _native.lib()
    return _native.HipEngine


def _t7_engine(cls, case, **over):
    cont = case["mode"] == 1
    in_dim, n_dist = (2 if case["env_id"] == 4 else 3, 2) if cont else ({3: 2, 5: 6}.get(case["env_id"], 4), case["num_actions"])
    e = cls(env_id=case["env_id"], mode=case["mode"], n_trees=case["n_games"], n_sims=case["n_sims"], c_uct=case["c_uct"],
            gamma=case["gamma"], epsilon=case["epsilon"], num_actions=case.get("num_actions", 0), c_pw=case.get("c_pw", 1.0),
            kappa=case.get("kappa", 0.5), v_target=case["v_target"], seed=case["seed"], tree_id_base=case["tree_id_base"],
            action_bound=case.get("action_bound", 2.0))
    e.set_weights(_capi.make_desc(in_dim, case["hidden"], n_dist, case["act"]),
                  O.make_weights(case["wseed"], in_dim, case["hidden"], n_dist, scale=case.get("wscale", 1.0)))
    kw = dict(max_episode_length=case["max_len"], deterministic=case.get("det", False), capacity_steps=case["n_steps"],
              final_selection=case.get("final_selection", "max_visit"), temperature=case.get("temperature", 1.0),
              agent_epsilon=case.get("agent_eps", 0.0))
    kw.update(over)
    e.selfplay_begin(**kw)
    if "first_roots" in case:   # (the T7 case's games start their first episode here instead of at the engine's reset state)
        fr = np.asarray(case["first_roots"], np.float64)
        e.upload_roots(fr[np.arange(case["n_games"]) % len(fr)])   # (larger batches of the same case: the roots repeat)
    return e


@pytest.mark.parametrize("which", ENGINES)
@pytest.mark.parametrize("name", T7)
def test_selfplay_matches_the_reference_run_loop(name, which):
    """T7 (tests/golden/gen_golden.py run_t7): the reference's ContinuousAgent.act / DiscreteAgent.act + Env.step + reset_mcts /
    mcts_forward (run_continuous.py:111-142, run_discrete.py:94-122) for several games over episode boundaries.  Per step and
    game: the replay row buffer.store received, the root the search started from and the count it carried; per game: finished
    episodes and their returns.  Integers (counts, episode statistics) and actions exactly; float32 row entries to float32
    rounding of values that agree to 1e-12 in float64."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"t7_selfplay_{name}.npz"))
    case = ast.literal_eval(str(z["case"]))
    e = _t7_engine(_engine_cls(which), case)
    G, n_steps = case["n_games"], case["n_steps"]
    So, K = e.s_obs, e.kmax
    roots = []
    for _ in range(n_steps):
        roots.append(e.selfplay_stats()[2].copy())
        e.selfplay_step()
    rows = e.selfplay_rows(clear=False).reshape(n_steps, G, -1)
    fsum, fcnt, state = e.selfplay_stats()
    e.close()
    np.testing.assert_allclose(np.stack(roots), z["root_before"], rtol=0, atol=1e-12)          # the roots every search started from
    want = z["rows"]
    np.testing.assert_array_equal(rows[..., So + K:So + 2 * K], want[..., So + K:So + 2 * K])    # visit counts
    np.testing.assert_array_equal(rows[..., So:So + K], want[..., So:So + K].astype(np.float32))  # root actions
    np.testing.assert_allclose(rows[..., :So], want[..., :So], rtol=6e-8, atol=1.2e-7)           # observation (float32 rounding of the float64 value)
    np.testing.assert_allclose(rows[..., So + 2 * K:], want[..., So + 2 * K:], rtol=2e-7, atol=1e-7)   # Q and the value target
    np.testing.assert_array_equal(fcnt, z["fcnt"])
    np.testing.assert_allclose(fsum, z["fsum"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(state, z["final_state"], rtol=0, atol=1e-12)
    # the action every step took shows in the next root (same episode) -- and, for CartPole, in the carried root count
    if case["mode"] == 0:
        for s_ in range(1, n_steps):
            for g in range(G):
                if z["carry_in"][s_, g] > 0:   # (one env step apart; Acrobot's velocities move by more than 1 in a 0.2 s step)
                    assert abs(z["root_before"][s_, g] - z["root_before"][s_ - 1, g]).max() < (8.0 if case["env_id"] == 5 else 1.0)


@pytest.mark.parametrize("which", ENGINES)
def test_replay_ring_overwrites_like_the_reference_buffer(which):
    """The FIFO rule of ReplayBuffer.store (buffers.py:75-82), pinned by the T6 golden (the reference's buffer of max_size 7
    through 17 stores): one game, ring of 7 steps, 17 steps played; after every step the ring's slots hold the steps the
    reference's slots hold, and size / insert_index agree."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t6_buffer.npz"))
    zc = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t7_selfplay_cartpole_sampled.npz"))
    case = dict(ast.literal_eval(str(zc["case"])), n_games=1)
    cls = _engine_cls(which)
    full = _t7_engine(cls, case, capacity_steps=17)                 # the same 17 steps without overwriting: row of step i
    ring = _t7_engine(cls, case, capacity_steps=7, fifo=True)
    for i in range(17):
        full.selfplay_step()
        ring.selfplay_step()
        all_rows = full.selfplay_rows(clear=False)
        got = ring.selfplay_rows(clear=False)
        slots = z["slots"][i]
        size, insert = int(slots[8]), int(slots[7])
        assert ring.selfplay_ring() == (size, insert, i + 1)
        assert got.shape[0] == size
        for j in range(size):
            np.testing.assert_array_equal(got[j], all_rows[int(slots[j])], err_msg=f"after store {i}: slot {j}")
    ring.selfplay_rows(clear=True)
    assert ring.selfplay_ring()[:2] == (0, 0)                        # ReplayBuffer.clear (buffers.py:56-60)
    stop = _t7_engine(cls, case, capacity_steps=2)
    stop.selfplay_step(); stop.selfplay_step()
    with pytest.raises(_capi.EngineError):
        stop.selfplay_step()                                         # AZG_RING_STOP: refuses instead of overwriting
    for e in (full, ring, stop):
        e.close()


def test_selfplay_config_errors():
    e = O.OracleEngine(env_id=0, mode=0, n_trees=2, n_sims=4, c_uct=1.5, gamma=1.0, num_actions=2)
    e.set_weights(_capi.make_desc(4, [64], 2, "relu"), O.make_weights(1, 4, [64], 2))
    with pytest.raises(_capi.EngineError):
        e.selfplay_begin(5, final_selection="max_value", temperature=0.5)     # Q-based sampling: temperature 1 only
    with pytest.raises(_capi.EngineError):
        e.selfplay_begin(5, temperature=0.0)
    with pytest.raises(_capi.EngineError):
        e.selfplay_begin(5, agent_epsilon=1.5)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", T7)
def test_selfplay_variants_hip_matches_oracle_bit_for_bit(name):
    """Every final-action rule of T7 on a larger ragged batch: HIP rows == oracle rows bit for bit."""
    from alphazero_gym_amd import _native
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"t7_selfplay_{name}.npz"))
    case = dict(ast.literal_eval(str(z["case"])), n_games=37, n_steps=12)
    out = []
    for cls in (_native.HipEngine, O.OracleEngine):
        e = _t7_engine(cls, case, capacity_steps=5, fifo=True)
        for _ in range(12):
            e.selfplay_step()
        out.append((e.selfplay_rows(clear=False), e.selfplay_stats(), e.selfplay_ring()))
        e.close()
    np.testing.assert_array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    for x, y in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(x, y)
    assert out[0][2] == out[1][2]


def _host_ring(ptr, shape, device):
    """The CPU oracle standing in for the engine: its "device" pointer is host memory (test double, injected from here)."""
    import ctypes
    import torch
    buf = (ctypes.c_float * (shape[0] * shape[1])).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.float32).reshape(shape))


@pytest.mark.parametrize("which", ENGINES)
def test_device_replay_samples_like_the_reference_buffer(which):
    """DeviceReplay = the ring + the reference's minibatch rule (buffers.py:84-123), pinned by the T6 golden: same shuffles under
    numpy seed 123, same batch sizes (3, 4 = last batch absorbs the remainder), same experiences in every batch, two epochs."""
    import torch
    from alphazero_gym_amd.agent.buffers import DeviceReplay
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t6_buffer.npz"))
    zc = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t7_selfplay_cartpole_sampled.npz"))
    case = dict(ast.literal_eval(str(zc["case"])), n_games=1)
    cls = _engine_cls(which)
    full = _t7_engine(cls, case, capacity_steps=17)
    ring = _t7_engine(cls, case, capacity_steps=7, fifo=True)
    for _ in range(17):
        full.selfplay_step()
        ring.selfplay_step()
    all_rows = full.selfplay_rows(clear=False)
    rep = DeviceReplay(ring, batch_size=3, **(dict(device="cpu", wrap=_host_ring) if which == "oracle" else {}))
    assert len(rep) == 7 and (rep.rows().is_cuda == (which == "hip"))
    np.testing.assert_array_equal(rep.rows().cpu().numpy(), ring.selfplay_rows(clear=False))      # zero-copy view == download
    np.random.seed(123)
    rep.reshuffle()
    got = []
    for epoch in range(2):
        for states, actions, counts, Qs, values in rep:
            assert states.shape[1] == 4 and actions.shape[1] == counts.shape[1] == Qs.shape[1] == 2 and values.ndim == 1
            rows = torch.cat([states, actions, counts, Qs, values[:, None]], 1).cpu().numpy()
            ids = [int(np.where((all_rows == r).all(1))[0][0]) for r in rows]                        # which step each row is
            got.append([epoch, len(ids)] + ids + [-1] * (8 - len(ids)))
    np.testing.assert_array_equal(np.array(got), z["batches"])
    # the view aliases the ring: another step shows up without re-wrapping
    before = rep.rows().clone()
    ring.selfplay_step()
    assert not torch.equal(before, rep.rows())
    # a second selfplay_begin replaces the ring: the view follows it instead of dangling
    ring.selfplay_begin(case["max_len"], capacity_steps=3, fifo=True)
    ring.selfplay_step()
    assert len(rep) == 1 and rep.ring.shape[0] == 3
    np.testing.assert_array_equal(rep.rows().cpu().numpy(), ring.selfplay_rows(clear=False))
    full.close(); ring.close()


@pytest.mark.gpu
def test_training_from_the_device_ring_without_a_host_copy():
    """DeviceSelfPlay.collect_device + train_on_rows with the policy on the GPU: rows stay device tensors from the ring to the
    optimiser step; same rows as the host download path."""
    import torch
    from alphazero_gym_amd import run
    from alphazero_gym_amd.agent.agents import ContinuousAgent
    cfg = run.CONTINUOUS_DEFAULTS
    policy = dict(cfg["policy"], hidden_dimensions=[64, 64], representation_dim=3, action_dim=1, action_bound=2.0)
    torch.manual_seed(0)
    agent = ContinuousAgent(policy_cfg=policy, mcts_cfg=dict(cfg["mcts"], n_rollouts=16, device="cuda:0"), loss_cfg=dict(run.LOSS_TUNED, device="cuda:0"),
                            optimizer_cfg=run.RMSPROP, device="cuda:0", **cfg["agent"])
    kw = dict(game="Pendulum-v1", n_games=48, n_rollouts=16, c_uct=0.05, max_episode_length=20, seed=3)
    sp_fifo = run.DeviceSelfPlay(agent.nn, capacity_steps=4, fifo=True, **kw)
    sp_host = run.DeviceSelfPlay(agent.nn, capacity_steps=6, **kw)
    rep = sp_fifo.replay(batch_size=32)
    for it in range(3):   # 6 steps per iteration through a ring of 4: wraps every time
        dev_rows = sp_fifo.collect_device(3, rep)
        host_rows = sp_host.collect(3)
        assert dev_rows.is_cuda and dev_rows.shape == (3 * 48, sp_fifo.engine.s_obs + 3 * sp_fifo.engine.kmax + 1)
        np.testing.assert_array_equal(dev_rows.cpu().numpy(), host_rows.numpy())
    w0 = [p.detach().clone() for p in agent.nn.parameters()]
    info = run.train_on_rows(agent, dev_rows, 3, sp_fifo.engine.kmax, batch_size=32)
    assert np.isfinite(info["loss"]) and any(not torch.equal(a, b) for a, b in zip(w0, agent.nn.parameters()))
    rep.reshuffle()
    n = 0
    for states, actions, counts, Qs, values in rep:
        assert states.is_cuda
        info = agent.update((states, actions, counts, Qs, values))
        n += 1
    assert n == (4 * 48) // 32 and np.isfinite(info["loss"])
    # the next self-play step runs with the trained weights
    from alphazero_gym_amd.search.mcts import _weights_version
    stale = sp_fifo.mcts._version
    sp_fifo.play(1)
    assert sp_fifo.mcts._version == _weights_version(agent.nn) != stale
