"""CPU BASELINE / ORACLE.  TEST INFRASTRUCTURE ONLY (tests/ and bench.py's cpu_baseline leg).

A Python object-tree restatement of the reference's search with the reference's per-simulation cost structure: one Python
object per node and per edge, the environment deep-copied and replayed from the root in every simulation, numpy UCT over a
list comprehension, and batch-1 torch-CPU forwards through the policy module (one for the new node's value, one -- with
torch.distributions built per call -- at every widening node).  It is what "the reference Python/CPU path" costs on a
machine where the reference's own files cannot run; the C oracle next to it is the
optimised CPU port.

Parity status: PINNED by tests/test_pytree_baseline.py against the T3 golden (the reference itself run with its torch policy
and the engine's noise: identical visit counts, Q within 1e-9).

Reference lines followed (alphazero/search/...):
  mcts.py:656-702  MCTSContinuous.search       -> search_continuous
  mcts.py:418-462  MCTSDiscrete.search         -> search_discrete
  mcts.py:704-741 / 464-493  selectionUCT      -> _select_continuous / _select_discrete
  states.py:252-275  NodeContinuous.check_pw   -> inline in _select_continuous
  mcts.py:602-654  add_value_estimate / add_pw_action -> _value / _widen
  mcts.py:385-416  MCTSDiscrete.evaluation     -> _evaluate_discrete
  mcts.py:241-267  backprop, states.py:97-112 Action.update -> _backup
  mcts.py:269-307  return_results              -> root_results
"""
import copy
import math

import numpy as np
import torch

PENDULUM_R_SCALE = 16.2736044   # mcts.py:20


class Edge:
    __slots__ = ("action", "parent", "W", "n", "Q", "child")

    def __init__(self, action, parent, q_init):
        self.action, self.parent, self.W, self.n, self.Q, self.child = action, parent, 0.0, 0, q_init, None


class Node:
    __slots__ = ("state", "r", "terminal", "parent_edge", "n", "V", "edges", "priors")

    def __init__(self, state, r, terminal, parent_edge):
        self.state, self.r, self.terminal, self.parent_edge = state, r, terminal, parent_edge
        self.n, self.V, self.edges, self.priors = 0, None, [], None


def _obs_tensor(state):
    return torch.from_numpy(np.asarray(state)[None,]).float()


def _value(policy, node):
    node.V = np.squeeze(policy.predict_V(_obs_tensor(node.state))) if not node.terminal else np.array(0.0)


def _widen(policy, node):
    action = policy.sample_action(_obs_tensor(node.state))
    node.edges.append(Edge(action, node, node.V))


def _select_continuous(policy, node, c_uct, c_pw, kappa):
    if math.ceil(c_pw * (node.n + 1) ** kappa) > len(node.edges):
        _widen(policy, node)
        return node.edges[-1]
    scores = np.array([e.Q + c_uct * (np.sqrt(node.n + 1) / (e.n + 1)) for e in node.edges])
    return node.edges[int(np.argmax(scores))]


def _backup(node, gamma):
    R = node.V
    while node.parent_edge is not None:
        R = node.r + gamma * R
        edge = node.parent_edge
        edge.n += 1
        edge.W += R
        edge.Q = edge.W / edge.n
        node = edge.parent
        node.n += 1


def search_continuous(policy, env, n_rollouts, c_uct, c_pw, kappa, gamma, reward_scale=PENDULUM_R_SCALE):
    root = Node(np.asarray(env._get_obs()), 0.0, False, None)
    _value(policy, root)
    _widen(policy, root)
    for _ in range(n_rollouts):
        node = root
        sim = copy.deepcopy(env)
        while not node.terminal:
            edge = _select_continuous(policy, node, c_uct, c_pw, kappa)
            obs, reward, done, _ = sim.step(edge.action)
            reward = reward / reward_scale
            if edge.child is not None:
                node = edge.child
                continue
            node = Node(np.squeeze(obs), reward, done, edge)
            edge.child = node
            _value(policy, node)
            break
        _backup(node, gamma)
    return root


def _evaluate_discrete(policy, node, num_actions):
    x = _obs_tensor(node.state)
    node.V = np.squeeze(policy.predict_V(x)).item() if not node.terminal else 0.0
    node.edges = [Edge(a, node, node.V) for a in range(num_actions)]
    node.priors = policy.predict_pi(x).flatten()


def _select_discrete(node, c_uct):
    scores = np.array([e.Q + prior * c_uct * (np.sqrt(node.n + 1) / (e.n + 1)) for e, prior in zip(node.edges, node.priors)])
    return node.edges[int(np.argmax(scores))]


def search_discrete(policy, env, n_rollouts, c_uct, gamma, num_actions=2):
    root = Node(np.array(env.state, dtype=np.float32), 0.0, False, None)
    _evaluate_discrete(policy, root, num_actions)
    for _ in range(n_rollouts):
        node = root
        sim = copy.deepcopy(env)
        while not node.terminal:
            edge = _select_discrete(node, c_uct)
            obs, reward, done, _ = sim.step(edge.action)
            if edge.child is not None:
                node = edge.child
                continue
            node = Node(obs, reward, done, edge)
            edge.child = node
            _evaluate_discrete(policy, node, num_actions)
            break
        _backup(node, gamma)
    return root


def root_results(root):
    """(actions, counts, Q, off-policy value target) at the root."""
    counts = np.array([e.n for e in root.edges])
    Q = np.array([float(np.asarray(e.Q).reshape(-1)[0]) for e in root.edges])
    actions = np.array([float(np.asarray(e.action).reshape(-1)[0]) for e in root.edges], dtype=np.float32)
    return actions, counts, Q, Q.max()


def _worker(args):
    """One process of the throughput measurement: whole searches until at least `n_trees` are done AND `seconds` have passed
    (start-up -- imports, building the policy -- is outside the clock); returns (simulations, seconds)."""
    import time
    kind, n_trees, n_rollouts, hidden, seed, seconds = args
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from alphazero_gym_amd.envs import CartPoleEnv, PendulumEnv
    from alphazero_gym_amd.network.policies import make_policy
    torch.set_num_threads(1)
    torch.manual_seed(34)
    rng = np.random.RandomState(seed)
    if kind == "pendulum":
        policy = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                             num_components=1, action_bound=2.0)
    else:
        policy = make_policy(representation_dim=4, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity="relu",
                             num_actions=2)
    t0 = time.perf_counter()
    done = 0
    while done < n_trees or time.perf_counter() - t0 < seconds:
        done += 1
        if kind == "pendulum":
            env = PendulumEnv(state=[rng.uniform(-np.pi, np.pi), rng.uniform(-1, 1)], version=1)
            search_continuous(policy, env, n_rollouts, 0.05, 1, 0.5, 1)
        else:
            env = CartPoleEnv(state=rng.uniform(-0.05, 0.05, 4))
            search_discrete(policy, env, n_rollouts, 1.5, 1)
    return done * n_rollouts, time.perf_counter() - t0


def throughput(kind="pendulum", n_rollouts=200, hidden=(256, 256), processes=1, trees_per_process=2, seconds=0.0, detail=False):
    """sims/s of `processes` worker processes x 1 torch thread.  Every worker searches at least `trees_per_process` trees and
    for at least `seconds` of its own clock (the workers run side by side for that time); the rate is total simulations /
    slowest worker's wall time.  detail: also return (total simulations, slowest worker's seconds)."""
    import multiprocessing as mp
    args = [(kind, trees_per_process, n_rollouts, list(hidden), 100 + i, float(seconds)) for i in range(processes)]
    if processes == 1:
        res = [_worker(args[0])]
    else:
        with mp.get_context("spawn").Pool(processes) as pool:
            res = pool.map(_worker, args)
    rate = sum(r[0] for r in res) / max(r[1] for r in res)
    return (rate, sum(r[0] for r in res), max(r[1] for r in res)) if detail else rate
