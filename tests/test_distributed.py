"""CPU, world sizes 2, 4 and 8 over gloo: the N>1 path (tree sharding by global id, replay all-gather, weight broadcast).
The engine is the oracle test double here; on the GPU box the same code runs over RCCL with the HIP engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from alphazero_gym_amd import _capi, distributed as D

B_TOTAL, N_SIMS = 10, 40
KW = dict(env_id=2, mode=1, n_sims=N_SIMS, c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, seed=34)


def _search(n_trees, base):
    e = O.OracleEngine(n_trees=n_trees, tree_id_base=base, **KW)
    e.set_weights(_capi.make_desc(3, [64, 64], 2, "elu"), O.make_weights(34, 3, [64, 64], 2))
    roots = e.synthetic_roots()
    e.search(roots)
    r = e.results()
    e.close()
    return roots, r


def _worker(rank, world, port, q):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    O.set_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    lo, hi = D.shard_range(B_TOTAL, rank, world)
    roots, r = _search(hi - lo, lo)
    rows = D.pack_replay_rows(roots, r["actions"], r["counts"], r["Q"], r["v_target"])
    # block lengths: computed locally from shard_range (world 2: equal shards -> one all_gather_into_tensor; world 4: uneven),
    # or exchanged first when the caller does not know the partition (world 8)
    allrows = D.gather_replay_rows(rows, total=B_TOTAL) if world < 8 else D.gather_replay_rows(rows)
    model = torch.nn.Linear(4, 3)
    with torch.no_grad():
        for p in model.parameters():
            p.fill_(float(rank + 1))
    D.broadcast_weights(model, src=0)
    ok = all(bool((p == 1.0).all()) for p in model.parameters())
    if rank == 0:
        q.put((allrows.numpy(), ok))
    else:
        q.put((None, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_every_tree_once():
    for n in (1, 7, 10, 4096, 32768):
        for w in (1, 2, 3, 4, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_rank_selfplay_equals_single_process(world):
    """SURVEY 8e on the CPU double: 10 games over 2, 4 and 8 ranks (4 and 8 do not divide 10: shards of 3/3/2/2 and 2/2/1/.../1
    games, eight different tree_id_base offsets) -- the gathered replay rows equal the single-process run's row for row."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in got)                       # weight broadcast reached every rank
    rows = [g for g, _ in got if g is not None][0]
    roots, r = _search(B_TOTAL, 0)                        # the same 10 games in one process
    want = D.pack_replay_rows(roots, r["actions"], r["counts"], r["Q"], r["v_target"]).numpy()
    np.testing.assert_array_equal(rows, want)             # per-tree results do not depend on the number of ranks
    s_, a_, c_, q_, v_ = D.unpack_replay_rows(torch.from_numpy(rows), 2, r["actions"].shape[1])
    np.testing.assert_array_equal(c_.sum(1), np.full(B_TOTAL, N_SIMS))


def _train_worker(rank, world, port, q, hip=False):
    """examples/selfplay_train.py in its multi-rank form (gloo here, RCCL on the GPU box), engine = the oracle test double
    (CPU suite) or the HIP engine with both ranks on GPU 0 (-m gpu)"""
    import importlib.util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0" if hip else str(rank))
    from alphazero_gym_amd import _native
    if hip:
        _native.lib()
    else:
        _native.HipEngine = O.OracleEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "selfplay_train.py")
    spec = importlib.util.spec_from_file_location("selfplay_train", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = mod.parse_args(["--game", "CartPole-v0", "--games", "8", "--n-rollouts", "8", "--iters", "3", "--steps-per-iter", "6",
                           "--train-rows", "64", "--batch-size", "16", "--hidden", "64", "64", "--device", "cpu"])
    captured = {}
    orig = mod.build_agent

    def build(*a, **k):
        agent, sd = orig(*a, **k)
        captured["agent"] = agent
        return agent, sd

    mod.build_agent = build
    orig_sp = mod.run.DeviceSelfPlay

    def selfplay(*a, **k):
        captured["sp"] = orig_sp(*a, **k)
        return captured["sp"]

    mod.run.DeviceSelfPlay = selfplay
    hist = mod.train(args, log=None)
    flat = torch.cat([p.detach().reshape(-1) for p in captured["agent"].nn.parameters()])
    # what the ENGINE evaluates with, not just what torch holds: the ordinary (non-forced) weight sync of the next collect(),
    # then the root value of the same states on every rank
    sp = captured["sp"]
    sp.mcts.sync_weights()
    probe = np.tile(np.array([[0.01, -0.02, 0.03, 0.04]]), (8, 1)) * (0.5 * np.arange(1, 9))[:, None]
    sp.engine.search(probe)
    v_engine, _ = sp.engine.root_eval()
    with torch.no_grad():
        v_torch = captured["agent"].nn(torch.from_numpy(probe.astype(np.float32)))[1].reshape(-1).numpy()
    q.put((rank, hist, float(flat.double().sum()), float(flat.abs().double().sum()), v_engine.tolist(), v_torch.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _gather_worker(rank, world, port, q):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    out = {}
    # (a) the training example's shape: every rank plays `games` games for `steps` steps, games % world != 0
    games, steps = 3, 5
    mine = torch.full((games * steps, 2), float(rank)) + torch.arange(games * steps, dtype=torch.float32)[:, None] / 100
    out["equal"] = D.gather_replay_rows(mine, counts=[mine.shape[0]] * world).numpy()
    # (b) a shard_range partition of fewer games than ranks: ranks 2.. hold nothing; rows per game stated, not inferred
    total, rpg = 2, 3
    lo, hi = D.shard_range(total, rank, world)
    blk = torch.arange(lo * rpg, hi * rpg, dtype=torch.float32)[:, None].repeat(1, 2)
    out["sparse"] = D.gather_replay_rows(blk, total=total, rows_per_game=rpg).numpy()
    # (c) ... and inferring rows_per_game from an empty block is refused (on the ranks that cannot know it), before any collective
    try:
        D.gather_replay_rows(blk, total=total) if hi == lo else None
        out["refused"] = hi != lo
    except ValueError as ex:
        out["refused"] = "rows_per_game" in str(ex)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_replay_rows_block_lengths_without_an_exchange():
    """`total` is the games of the whole job; a rank without games contributes no rows when rows_per_game is given; equal
    blocks of games x steps rows (the training example, games % world != 0) go through `counts`."""
    world = 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        eq = got[r]["equal"]
        assert eq.shape == (world * 15, 2)
        for k in range(world):
            np.testing.assert_array_equal(eq[15 * k:15 * (k + 1), 0], np.float32(k) + np.arange(15, dtype=np.float32) / 100)
        np.testing.assert_array_equal(got[r]["sparse"][:, 0], np.arange(6, dtype=np.float32))
        assert got[r]["refused"]


@pytest.mark.parametrize("hip", [False, pytest.param(True, marks=pytest.mark.gpu)], ids=["oracle_double", "hip"])
def test_two_rank_training_loop_keeps_weights_in_sync(hip):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, hip)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, h0, s0, a0, ve0, vt0), (_, h1, s1, a1, ve1, vt1) = got
    assert s0 == s1 and a0 == a1 and np.isfinite(s0)                     # rank 0 trained, rank 1 received the same weights
    assert ve0 == ve1                                                    # ... and both ENGINES search with them
    np.testing.assert_allclose(ve1, vt1, atol=1e-5, rtol=1e-5)           # (the broadcast must reach the engine of every rank)
    assert len(h0) == 3 and h0[-1]["env_steps"] == 3 * 6 * 8 * 2         # both ranks' games counted
    keys = ("iter", "episodes_finished", "mean_return", "env_steps")
    assert [[h[k] for k in keys] for h in h0] == [[h[k] for k in keys] for h in h1]   # all-reduced episode statistics agree
    assert sum(h["episodes_finished"] for h in h0) > 0
