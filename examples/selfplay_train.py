#!/usr/bin/env python3
"""Batched self-play + training, end to end on one GPU: the scaled-out form of run_continuous.py:111-142 /
run_discrete.py:94-122.  B games live on the device (azg_selfplay_*: search, final action, env step, resets, replay rows);
every iteration downloads the new replay rows, runs the reference's loss / optimiser step in PyTorch on a random subset,
and re-syncs the weights into the engine.

    python examples/selfplay_train.py --game CartPole-v0 --games 512 --n-rollouts 32 --iters 30
    python examples/selfplay_train.py --game Pendulum-v1 --games 512 --n-rollouts 50 --iters 40

Prints the mean return of the episodes finished in each iteration (one JSON line per iteration).

Multi-GPU (one process per GPU, RCCL):  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
    examples/selfplay_train.py ...    Every rank plays --games games of its own (global game ids rank*games ...), the replay
rows are all-gathered, rank 0 runs the optimiser step and broadcasts the weights (alphazero_gym_amd/distributed.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_gym_amd import distributed as D, run  # noqa: E402
from alphazero_gym_amd.agent.agents import ContinuousAgent, DiscreteAgent  # noqa: E402


def build_agent(game, hidden, n_rollouts, device, lr):
    opt = dict(run.RMSPROP, lr=lr)
    loss = dict(run.LOSS_TUNED, device=device)
    if game.lower().startswith("pendulum"):
        cfg = run.CONTINUOUS_DEFAULTS
        policy = dict(cfg["policy"], hidden_dimensions=hidden, representation_dim=3, action_dim=1, action_bound=2.0)
        mcts = dict(cfg["mcts"], n_rollouts=n_rollouts, device=device)
        return ContinuousAgent(policy_cfg=policy, mcts_cfg=mcts, loss_cfg=loss, optimizer_cfg=opt, device=device, **cfg["agent"]), 3
    cfg = run.DISCRETE_DEFAULTS
    g = game.lower()   # MountainCar-v0: two observations, three actions; Acrobot-v1: six observations, three actions
    obs_dim, n_act = (2, 3) if g.startswith("mountaincar") else ((6, 3) if g.startswith("acrobot") else (4, 2))
    policy = dict(cfg["policy"], hidden_dimensions=hidden, representation_dim=obs_dim, action_dim=1, num_actions=n_act)
    mcts = dict(cfg["mcts"], n_rollouts=n_rollouts, device=device, num_actions=n_act)
    return DiscreteAgent(policy_cfg=policy, mcts_cfg=mcts, loss_cfg=loss, optimizer_cfg=opt, device=device, **cfg["agent"]), obs_dim


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="CartPole-v0")
    ap.add_argument("--games", type=int, default=512)
    ap.add_argument("--n-rollouts", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--steps-per-iter", type=int, default=20)
    ap.add_argument("--train-rows", type=int, default=2048, help="replay rows sampled per iteration")
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--hidden", type=int, nargs="+", default=[128, 128])
    ap.add_argument("--max-episode-length", type=int, default=200)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=34)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    return ap.parse_args(argv)


def train(a, log=print):
    """Runs the loop; returns the per-iteration records."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    torch.manual_seed(a.seed)   # same initial weights on every rank
    agent, state_dim = build_agent(a.game, a.hidden, a.n_rollouts, a.device, a.lr)
    continuous = state_dim == 3
    m = agent.mcts
    sp = run.DeviceSelfPlay(agent.nn, game=a.game, n_games=a.games, n_rollouts=a.n_rollouts, c_uct=m.c_uct, gamma=m.gamma,
                            epsilon=m.epsilon, c_pw=getattr(m, "c_pw", 1.0), kappa=getattr(m, "kappa", 0.5),
                            max_episode_length=a.max_episode_length, capacity_steps=a.steps_per_iter, seed=a.seed, rank=rank,
                            device_id=torch.cuda.current_device() if a.device.startswith("cuda") else 0)
    K = sp.engine.kmax if continuous else 2
    rng = np.random.RandomState(a.seed)
    fs0, fc0 = 0.0, 0
    t0 = time.time()
    history = []
    # with the policy on the GPU the replay rows never leave HBM: the ring is wrapped as a torch tensor (zero copy), the
    # all-gather runs device to device over xGMI, and the minibatches are gathered on the GPU
    on_gpu = a.device.startswith("cuda")
    replay = sp.replay(a.batch_size) if on_gpu else None
    for it in range(a.iters):
        rows = sp.collect_device(a.steps_per_iter, replay) if on_gpu else sp.collect(a.steps_per_iter)
        if world > 1:
            # every rank plays a.games games for steps_per_iter steps: equal blocks, known without a length exchange (whatever a.games
            # and the world size are: the job as a whole plays a.games * world games)
            rows = D.gather_replay_rows(rows, counts=[rows.shape[0]] * world)
        info = {"loss": 0.0}
        pick = rng.choice(rows.shape[0], size=min(a.train_rows, rows.shape[0]), replace=False)
        if rank == 0:
            info = run.train_on_rows(agent, rows[torch.from_numpy(pick).to(rows.device)], state_dim, K, batch_size=a.batch_size, shuffle_seed=it)
        if world > 1:
            D.broadcast_weights(agent.nn, src=0)
        fsum, fcnt, _ = sp.engine.selfplay_stats()
        stats = torch.tensor([float(fsum.sum()), float(fcnt.sum())], dtype=torch.float64)
        if world > 1:
            stats = stats.to(a.device if a.device.startswith("cuda") else "cpu")
            dist.all_reduce(stats)
        fs, fc = float(stats[0]), int(stats[1])
        mean_ret = (fs - fs0) / max(fc - fc0, 1)
        n_batches = max(1, len(pick) // a.batch_size)
        history.append({"iter": it, "episodes_finished": fc - fc0, "mean_return": round(mean_ret, 2),
                        "loss": round(info["loss"] / n_batches, 4), "env_steps": (it + 1) * a.steps_per_iter * a.games * world,
                        "elapsed_s": round(time.time() - t0, 1)})
        if log and rank == 0:
            log(json.dumps(history[-1]), flush=True)
        fs0, fc0 = fs, fc
    return history


def main():
    a = parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        if a.device.startswith("cuda"):
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group("gloo")
    train(a)


if __name__ == "__main__":
    main()
