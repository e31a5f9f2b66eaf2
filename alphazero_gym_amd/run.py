"""Training drivers with the reference's loop shape, without hydra / wandb / gym.

``run_continuous_agent`` / ``run_discrete_agent`` mirror run_continuous.py:15-165 / run_discrete.py:16-146: one game, one
tree, ``act -> buffer.store -> Env.step -> reset_mcts | mcts_forward`` per step, ``agent.train(buffer)`` per episode.
``BatchedSelfPlay`` is the scaled-out form the engine is built for: B games per GPU advance in lock step, one search
launch per environment step, replay rows gathered across ranks, weights broadcast after the optimiser step.

Default hyper-parameters are the reference's (config/*.yaml), except that the continuous policy uses
one squashed-Normal component (num_components: 1) instead of the 2-component GMM.
"""
import argparse
import copy
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import _capi, distributed as D
from .agent.agents import ContinuousAgent, DiscreteAgent
from .agent.buffers import DeviceReplay, ReplayBuffer
from .envs import VecCartPole, VecMountainCarContinuous, VecPendulum, make_game
from .helpers import check_space, stable_normalizer
from .search.mcts import BatchedMCTS

LOSS_TUNED = dict(_target_="alphazero_gym_amd.agent.losses.A0CLossTuned", action_dim=1, alpha_init=1.0, lr=0.001, tau=0.1,
                  policy_coeff=0.1, value_coeff=1.0, reduction="mean", grad_clip=0, device="cpu")
RMSPROP = dict(_target_="torch.optim.RMSprop", lr=0.001, alpha=0.9, eps=1e-10, weight_decay=0, momentum=0)

CONTINUOUS_DEFAULTS = dict(
    game="Pendulum-v0", seed=34, num_train_episodes=45, max_episode_length=200, device="cpu",
    buffer=dict(max_size=3000, batch_size=32),
    policy=dict(_target_="alphazero_gym_amd.network.policies.make_policy", distribution="normal", num_components=1,
                hidden_dimensions=[128, 128, 128], nonlinearity="elu", layernorm=False, log_param_min=-5, log_param_max=2),
    mcts=dict(_target_="alphazero_gym_amd.search.mcts.MCTSContinuous", n_rollouts=25, c_pw=1, kappa=0.5, c_uct=0.05, gamma=1, epsilon=0,
              V_target_policy="off_policy", root_state=None),
    loss=LOSS_TUNED, optimizer=RMSPROP,
    agent=dict(final_selection="max_visit", epsilon=0, train_epochs=1, grad_clip=0),
)
DISCRETE_DEFAULTS = dict(
    game="CartPole-v0", seed=34, num_train_episodes=200, max_episode_length=200, device="cpu",
    buffer=dict(max_size=1000, batch_size=32),
    policy=dict(_target_="alphazero_gym_amd.network.policies.make_policy", distribution="discrete", hidden_dimensions=[128, 128],
                nonlinearity="relu", layernorm=False),
    mcts=dict(_target_="alphazero_gym_amd.search.mcts.MCTSDiscrete", n_rollouts=8, c_uct=1.5, gamma=1, epsilon=0.1,
              V_target_policy="off_policy", root_state=None),
    loss=LOSS_TUNED, optimizer=RMSPROP,
    agent=dict(final_selection="max_visits", temperature=1.0, train_epochs=1, grad_clip=0),
)


def _merge(base: dict, over: Optional[dict]) -> dict:
    out = copy.deepcopy(base)
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = v
    return out


def run_continuous_agent(cfg: Optional[dict] = None, log: Optional[Callable[[Dict, int], None]] = None) -> List[float]:
    """run_continuous.py:15-165"""
    cfg = _merge(CONTINUOUS_DEFAULTS, cfg)
    Env = make_game(cfg["game"])
    np.random.seed(cfg["seed"])
    Env.seed(cfg["seed"])
    buffer = ReplayBuffer(**cfg["buffer"])
    state_dim, _ = check_space(Env.observation_space)
    action_dim, discrete = check_space(Env.action_space)
    assert not discrete, "Using continuous agent for a discrete action space!"
    policy = dict(cfg["policy"], representation_dim=state_dim[0], action_dim=action_dim[0], action_bound=float(Env.action_space.high[0]))
    agent = ContinuousAgent(policy_cfg=policy, mcts_cfg=dict(cfg["mcts"], device=cfg["device"]), loss_cfg=cfg["loss"],
                            optimizer_cfg=cfg["optimizer"], device=cfg["device"], **cfg["agent"])
    returns = []
    for ep in range(cfg["num_train_episodes"]):
        state = Env.reset()
        R = 0.0
        agent.reset_mcts(root_state=state)
        for t in range(cfg["max_episode_length"]):
            action, s, actions, counts, Qs, V = agent.act(Env=Env)
            buffer.store((s, actions, counts, Qs, V))
            state, step_reward, terminal, _ = Env.step(action)
            R += float(np.asarray(step_reward).reshape(-1)[0])
            if terminal or t == cfg["max_episode_length"] - 1:
                break
            agent.reset_mcts(root_state=state)   # the continuous tree cannot be reused
        returns.append(R)
        info = agent.train(buffer)
        info["Episode reward"] = R
        if log:
            log(dict(info), ep)
    return returns


def run_discrete_agent(cfg: Optional[dict] = None, log: Optional[Callable[[Dict, int], None]] = None) -> List[float]:
    """run_discrete.py:16-146"""
    cfg = _merge(DISCRETE_DEFAULTS, cfg)
    Env = make_game(cfg["game"])
    np.random.seed(cfg["seed"])
    Env.seed(cfg["seed"])
    buffer = ReplayBuffer(**cfg["buffer"])
    state_dim, _ = check_space(Env.observation_space)
    action_dim, discrete = check_space(Env.action_space)
    assert discrete, "Can't use discrete agent for continuous action spaces!"
    policy = dict(cfg["policy"], representation_dim=state_dim[0], action_dim=1, num_actions=action_dim[0])
    agent = DiscreteAgent(policy_cfg=policy, mcts_cfg=dict(cfg["mcts"], device=cfg["device"], num_actions=action_dim[0]),
                          loss_cfg=cfg["loss"], optimizer_cfg=cfg["optimizer"], device=cfg["device"], **cfg["agent"])
    returns = []
    for ep in range(cfg["num_train_episodes"]):
        state = Env.reset()
        R = 0.0
        agent.reset_mcts(root_state=state)
        for t in range(cfg["max_episode_length"]):
            action, s, actions, counts, Qs, V = agent.act(Env=Env, deterministic=False)
            buffer.store((s, actions, counts, Qs, V))
            new_state, step_reward, terminal, _ = Env.step(action)
            R += step_reward
            if terminal or t == cfg["max_episode_length"] - 1:
                break
            agent.mcts_forward(action, new_state)
        returns.append(R)
        info = agent.train(buffer)
        info["Episode reward"] = R
        if log:
            log(dict(info), ep)
    return returns


class BatchedSelfPlay:
    """B games per process in lock step: one search launch per environment step.
    Final action rules are the agents': continuous -> most visited root action (first index on ties),
    discrete -> sampled in proportion to (counts/max)^temperature."""

    def __init__(self, policy, *, game: str, n_games: int, n_rollouts: int, c_uct: float, gamma: float = 1.0, epsilon: float = 0.0,
                 c_pw: float = 1.0, kappa: float = 0.5, V_target_policy: str = "off_policy", max_episode_length: int = 200,
                 temperature: float = 1.0, seed: int = 34, rank: int = 0, world: int = 1, device_id: int = 0):
        self.policy = policy
        self.continuous = game.lower().startswith(("pendulum", "mountaincarcontinuous"))
        self.n = n_games
        self.max_len = max_episode_length
        self.temperature = temperature
        self.rng = np.random.RandomState(seed + 1000 * rank)
        base = rank * n_games
        if self.continuous:
            if game.lower().startswith("mountaincarcontinuous"):
                self.env = VecMountainCarContinuous(n_games, seed=seed + rank)
            else:
                self.env = VecPendulum(n_games, version=0 if game.endswith("v0") else 1, seed=seed + rank)
            self.mcts = BatchedMCTS(policy, env_id=self.env.azg_env_id, mode=_capi.MODE_CONTINUOUS, n_trees=n_games, n_rollouts=n_rollouts,
                                    c_uct=c_uct, gamma=gamma, epsilon=epsilon, c_pw=c_pw, kappa=kappa, V_target_policy=V_target_policy,
                                    action_bound=float(policy.action_bound), seed=seed, tree_id_base=base, device_id=device_id)
        else:
            if not game.lower().startswith("cartpole"):
                # (this host-stepped loop has vectorised numpy envs for CartPole, Pendulum and MountainCarContinuous only)
                raise NotImplementedError(f"BatchedSelfPlay steps CartPole, Pendulum and MountainCarContinuous on the host; {game} plays "
                                          "its own dynamics on the device: use DeviceSelfPlay")
            self.env = VecCartPole(n_games, seed=seed + rank)
            self.mcts = BatchedMCTS(policy, env_id=_capi.ENV_CARTPOLE, mode=_capi.MODE_DISCRETE, n_trees=n_games, n_rollouts=n_rollouts,
                                    c_uct=c_uct, gamma=gamma, epsilon=epsilon, num_actions=policy.num_actions,
                                    V_target_policy=V_target_policy, seed=seed, tree_id_base=base, device_id=device_id)
        self.t = np.zeros(n_games, np.int64)
        self.ep_return = np.zeros(n_games)
        self.finished_returns: List[float] = []

    def step(self):
        """One environment step of all games; returns the replay rows (obs, actions, counts, Q, V_target) of this step."""
        obs = self.env.obs()
        self.mcts.search(self.env.state)
        r = self.mcts.results()
        K = int(r["n_children"].max())
        counts, Q, actions = r["counts"][:, :K], r["Q"][:, :K], r["actions"][:, :K]
        if self.continuous:
            act = actions[np.arange(self.n), counts.argmax(1)]
        else:
            pi = np.stack([stable_normalizer(c.astype(np.float64), self.temperature) for c in counts])
            act = np.array([self.rng.choice(K, p=p) for p in pi])
        reward, done = self.env.step(act)
        self.ep_return += reward
        self.t += 1
        over = done | (self.t >= self.max_len)
        if over.any():
            self.finished_returns += self.ep_return[over].tolist()
            self.ep_return[over] = 0.0
            self.t[over] = 0
            self.env.reset(over)
        return obs, actions, counts, Q, r["v_target"]

    def collect(self, n_steps: int) -> torch.Tensor:
        """Play n_steps and return this rank's replay rows packed as float32 [n_steps * B, row]."""
        rows = [D.pack_replay_rows(*self.step()) for _ in range(n_steps)]
        return torch.cat(rows, 0)


class DeviceSelfPlay:
    """The same loop with everything but the optimiser on the GPU (azg_selfplay_* in include/azgym.h): games, final action
    rule, env step, episode resets and the replay ring live on the device; the host only downloads rows to train on.
    The agents' final action rules run on the device too: ``final_selection`` "max_visit" / "max_value", discrete
    ``temperature`` and ``deterministic``, continuous ``agent_epsilon`` (agents.py:294-301, 524-535; include/azgym.h).
    ``fifo=True`` turns the replay ring into the reference's overwrite-the-oldest buffer (buffers.py:75-82)."""

    def __init__(self, policy, *, game: str, n_games: int, n_rollouts: int, c_uct: float, gamma: float = 1.0, epsilon: float = 0.0,
                 c_pw: float = 1.0, kappa: float = 0.5, V_target_policy: str = "off_policy", max_episode_length: int = 200,
                 deterministic: bool = False, capacity_steps: int = 64, seed: int = 34, rank: int = 0, device_id: int = 0,
                 final_selection: str = "max_visit", temperature: float = 1.0, agent_epsilon: float = 0.0, fifo: bool = False):
        self.continuous = game.lower().startswith(("pendulum", "mountaincarcontinuous"))
        if self.continuous:
            if game.lower().startswith("mountaincarcontinuous"):
                env_id = _capi.ENV_MOUNTAINCAR_CONT
            else:
                env_id = _capi.ENV_PENDULUM_V0 if game.endswith("v0") else _capi.ENV_PENDULUM_V1
            self.mcts = BatchedMCTS(policy, env_id=env_id, mode=_capi.MODE_CONTINUOUS, n_trees=n_games, n_rollouts=n_rollouts, c_uct=c_uct,
                                    gamma=gamma, epsilon=epsilon, c_pw=c_pw, kappa=kappa, V_target_policy=V_target_policy,
                                    action_bound=float(policy.action_bound), seed=seed, tree_id_base=rank * n_games, device_id=device_id)
        else:
            g = game.lower()
            env_id = _capi.ENV_MOUNTAINCAR if g.startswith("mountaincar") else (_capi.ENV_ACROBOT if g.startswith("acrobot") else _capi.ENV_CARTPOLE)
            self.mcts = BatchedMCTS(policy, env_id=env_id, mode=_capi.MODE_DISCRETE, n_trees=n_games, n_rollouts=n_rollouts,
                                    c_uct=c_uct, gamma=gamma, epsilon=epsilon, num_actions=policy.num_actions,
                                    V_target_policy=V_target_policy, seed=seed, tree_id_base=rank * n_games, device_id=device_id)
        self.engine = self.mcts.engine
        self.capacity = capacity_steps
        self.fifo = fifo
        self.engine.selfplay_begin(max_episode_length, deterministic, capacity_steps, final_selection=final_selection,
                                   temperature=temperature, agent_epsilon=agent_epsilon, fifo=fifo)

    def play(self, n_steps: int) -> None:
        """n_steps self-play steps of every game (asynchronous: launches only); rows accumulate in the device ring."""
        assert self.fifo or n_steps <= self.capacity
        self.mcts.sync_weights()
        for _ in range(n_steps):
            self.engine.selfplay_step()

    def collect(self, n_steps: int) -> torch.Tensor:
        """Play n_steps (<= capacity) and return this rank's replay rows as a host float32 tensor [n_steps * B, row]."""
        assert n_steps <= self.capacity
        self.play(n_steps)
        return torch.from_numpy(self.engine.selfplay_rows(clear=True).copy())

    def replay(self, batch_size: int, device=None) -> DeviceReplay:
        """The device ring as a replay buffer with the reference's sampling rules; minibatches are device tensors."""
        return DeviceReplay(self.engine, batch_size, device=device)

    def collect_device(self, n_steps: int, replay: DeviceReplay) -> torch.Tensor:
        """Play n_steps and return this rank's NEW rows as a device tensor [n_steps * B, row] (a copy in HBM; the ring is
        cleared unless it runs in FIFO mode, where it keeps accumulating)."""
        if not self.fifo:
            assert n_steps <= self.capacity
            self.play(n_steps)
            rows = replay.rows().clone()
            self.engine.selfplay_clear()
            return rows
        assert n_steps <= self.capacity, "the ring keeps its newest capacity_steps steps: older ones would already be overwritten"
        before = self.engine.selfplay_ring()
        self.play(n_steps)
        size, insert, _ = self.engine.selfplay_ring()
        B = self.engine.n_trees
        ring = replay.rows().reshape(size, B, -1)
        # the n_steps newest slots, oldest first: while filling they are the tail; once full they end just before insert_index
        if before[0] + n_steps <= self.capacity:
            new = ring[before[0]:before[0] + n_steps]
        else:
            idx = [(insert - n_steps + i) % size for i in range(n_steps)]
            new = ring[torch.as_tensor(idx, device=ring.device)]
        return new.reshape(n_steps * B, -1).clone()

    def mean_finished_return(self) -> float:
        fsum, fcnt, _ = self.engine.selfplay_stats()
        return float(fsum.sum() / max(int(fcnt.sum()), 1))


def train_on_rows(agent, rows: torch.Tensor, state_dim: int, K: int, batch_size: int = 32, shuffle_seed: int = 0) -> Dict[str, float]:
    """One epoch of minibatch updates (agent.update) over replay rows gathered from all ranks; the last batch absorbs the
    remainder like ReplayBuffer.__next__."""
    on_device = rows.is_cuda
    if on_device:   # minibatches are gathered on the GPU, nothing goes through the host
        s, a, c, q, v = (rows[:, :state_dim], rows[:, state_dim:state_dim + K], rows[:, state_dim + K:state_dim + 2 * K],
                         rows[:, state_dim + 2 * K:state_dim + 3 * K], rows[:, -1])
    else:
        s, a, c, q, v = D.unpack_replay_rows(rows, state_dim, K)
    n = s.shape[0]
    order = np.random.RandomState(shuffle_seed).permutation(n)
    sums: Dict[str, float] = {}
    i = 0
    while i < n:
        j = n if i + 2 * batch_size > n else i + batch_size
        idx = order[i:j]
        if on_device:
            ti = torch.from_numpy(idx).to(rows.device)
            info = agent.update((s[ti], a[ti], c[ti], q[ti], v[ti]))
        else:
            info = agent.update((s[idx].copy(), a[idx].copy(), c[idx].copy(), q[idx].copy(), v[idx].astype(np.float64)))
        for k_, val in info.items():
            sums[k_] = sums.get(k_, 0.0) + val
        i = j
    return sums


def main():
    ap = argparse.ArgumentParser(description="single-game drivers with the reference's loop shape")
    ap.add_argument("kind", choices=["continuous", "discrete"])
    ap.add_argument("--episodes", type=int, default=None)
    ap.add_argument("--n-rollouts", type=int, default=None)
    a = ap.parse_args()
    over = {}
    if a.episodes is not None:
        over["num_train_episodes"] = a.episodes
    if a.n_rollouts is not None:
        over["mcts"] = {"n_rollouts": a.n_rollouts}
    fn = run_continuous_agent if a.kind == "continuous" else run_discrete_agent
    rets = fn(over, log=lambda info, ep: print(ep, {k: round(float(v), 4) for k, v in info.items()}))
    print("episode returns:", np.round(rets, 2).tolist())


if __name__ == "__main__":
    main()
