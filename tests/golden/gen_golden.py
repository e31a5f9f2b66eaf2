#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE (build container only).

    python tests/golden/gen_golden.py            # needs /root/reference; writes tests/golden/*.npz
    python tests/golden/gen_golden.py t7         # only the self-play tier
    python tests/golden/gen_golden.py t1 NAME... # only the named T1 cases
    python tests/golden/gen_golden.py scale      # only t3_scale.npz (T3 on 1024 config-C, 1024 config-B, 64 config-E roots; ~5 min on 8 cores)
    python tests/golden/gen_golden.py wide       # only t2_mlp_wide.npz (4x1024, 2x512), t3_full_rollouts.npz (n_rollouts = 200), t5_update.npz

The reference (timoklein/alphazero-gym) is imported unmodified from /root/reference.  `gym`, `hydra`
and `omegaconf` are not installed in this image; the reference's hot path only needs their names for
type hints / base classes, so three empty stand-in modules are created in a temp dir.
Nothing of the reference's source is copied: the fixtures hold inputs and outputs only.

Tiers (SURVEY.md 7 "hard parts"):
  T1  tree logic.   The reference's MCTS*.search runs with a duck-typed model whose predict_V /
      predict_pi / sample_action call the C oracle's MLP (so network outputs are bit-identical to
      the engine's arithmetic) and with the engine's Philox draws injected for the squashed-Normal
      noise and epsilon-greedy.  Whatever differs afterwards is tree logic.  Ties in argmax are
      asserted absent (the reference breaks them randomly, the engine by lowest index).
  T2  MLP.          The reference's make_policy networks (torch) with the same synthetic weights:
      predict_V / predict_pi / forward outputs for a batch of observations (tolerance 1e-5).
  T3  end to end.   The reference with its real torch policy, torch.normal patched to the engine's
      noise: visit counts / Q for whole searches (near-tie flips from ~1e-7 MLP differences possible).
  T4  agent.        DiscreteAgent.act / ContinuousAgent.act return tuples (shapes, dtypes, final action).
  T5  training.     get_train_data and the losses on fixed batches.
  T6  replay.       The reference's ReplayBuffer through wrap-around and two epochs of minibatches.
  T7  self-play.    The reference's run loops (run_continuous.py:111-142, run_discrete.py:94-122: act -> buffer.store ->
      Env.step -> reset_mcts | mcts_forward, episode ends and resets) driven with the reference's own agents for several
      games over episode boundaries; pins azo_selfplay_* / azg_selfplay_* (replay rows, final actions, returns, carried roots).
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REFERENCE = "/root/reference"


def _install_stubs():
    d = tempfile.mkdtemp(prefix="azg_stubs_")
    os.makedirs(os.path.join(d, "gym"))
    with open(os.path.join(d, "gym", "__init__.py"), "w") as f:
        f.write("class Env: pass\nclass Wrapper(Env): pass\nfrom . import spaces\n")
    with open(os.path.join(d, "gym", "spaces.py"), "w") as f:
        f.write("class Box: pass\nclass Discrete: pass\n")
    os.makedirs(os.path.join(d, "hydra"))
    with open(os.path.join(d, "hydra", "__init__.py"), "w") as f:
        f.write("class utils:\n    call = staticmethod(lambda *a, **k: None)\n    instantiate = staticmethod(lambda *a, **k: None)\n"
                "def main(*a, **k):\n    return lambda f: f\n")
    os.makedirs(os.path.join(d, "omegaconf"))
    with open(os.path.join(d, "omegaconf", "__init__.py"), "w") as f:
        f.write("")
    with open(os.path.join(d, "omegaconf", "dictconfig.py"), "w") as f:
        f.write("class DictConfig(dict): pass\n")
    sys.path.insert(0, d)
    sys.path.insert(0, REFERENCE)


_install_stubs()

import torch  # noqa: E402

import alphazero.search.mcts as RM  # noqa: E402  (the reference)
import alphazero.search.states as RS  # noqa: E402
from alphazero.network.policies import make_policy  # noqa: E402

import oracle_lib as O  # noqa: E402
from alphazero_gym_amd import _capi  # noqa: E402
from alphazero_gym_amd.envs import AcrobotEnv, CartPoleEnv, MountainCarContinuousEnv, MountainCarEnv, PendulumEnv  # noqa: E402

torch.set_num_threads(1)
TIES = {"n": 0}


def argmax_first(x):
    x = x.flatten()
    w = np.where(x == np.max(x))[0]
    if len(w) > 1:
        TIES["n"] += 1
    return w[0]


RM.argmax = argmax_first


class EngineRandom:
    """Stands in for the `random` module inside alphazero.search.mcts: epsilon-greedy draws (mcts.py:190-192)."""

    def __init__(self, seed, tree, search):
        self.seed, self.tree, self.search, self.draw = seed, tree, search, 0
        self._r = 0

    def random(self):
        u, r = O.eps_draw(self.seed, self.tree, self.search, self.draw)
        self.draw += 1
        self._r = r
        return u

    def randint(self, a, b):
        return a + self._r % (b - a + 1)

    def choice(self, seq):
        return seq[0]


class OracleModel:
    """Duck-typed model (mcts.py:407, 416, 620, 652) backed by the C oracle's MLP + the engine's Philox noise."""

    def __init__(self, eng, seed, tree, search, bound):
        self.eng, self.seed, self.tree, self.search, self.bound = eng, seed, tree, search, bound
        self.n_sampled = 0

    def predict_V(self, x):
        v, _, _ = self.eng.mlp_eval(x.numpy())
        return v.reshape(1, 1)

    def predict_pi(self, x):
        _, d, _ = self.eng.mlp_eval(x.numpy())
        return d.reshape(1, -1)

    def sample_action(self, x):
        _, d, _ = self.eng.mlp_eval(x.numpy())
        self.n_sampled += 1
        eps = O.normal(self.seed, self.tree, self.search, self.n_sampled)  # draw index = record id of the new edge
        return np.array([[O.sample_action(d[0, 0], d[0, 1], eps, self.bound)]], dtype=np.float32)


def number_actions():
    """Give every reference Action object its creation index == the engine's record id."""
    counter = {"n": 0}
    oc, od = RS.ActionContinuous.__init__, RS.ActionDiscrete.__init__

    def ic(self, action, parent_node, Q_init):
        oc(self, action, parent_node, Q_init)
        counter["n"] += 1
        self._rec = counter["n"]

    def idd(self, action, parent_node, Q_init):
        od(self, action, parent_node, Q_init)
        if parent_node.terminal:   # the engine creates no edges under terminal nodes (never observable)
            self._rec = -1
        else:
            counter["n"] += 1
            self._rec = counter["n"]

    RS.ActionContinuous.__init__ = ic
    RS.ActionDiscrete.__init__ = idd
    RM.ActionContinuous.__init__ = ic
    RM.ActionDiscrete.__init__ = idd
    return counter


COUNTER = number_actions()


def dump_reference_tree(root, R):
    d = {
        "parent": np.zeros(R, np.int32), "edge_n": np.zeros(R, np.int32), "edge_W": np.zeros(R, np.float64),
        "edge_Q": np.zeros(R, np.float64), "edge_action": np.zeros(R, np.float32), "node_n": np.zeros(R, np.int32),
        "node_r": np.zeros(R, np.float64), "node_V": np.zeros(R, np.float32), "node_flags": np.zeros(R, np.uint8),
    }
    d["parent"][0] = -1
    nrec = 1

    def visit(node, rec):
        nonlocal nrec
        d["node_n"][rec] = node.n
        d["node_r"][rec] = float(np.asarray(node.r).reshape(-1)[0])
        d["node_V"][rec] = np.float32(np.asarray(node.V).reshape(-1)[0])
        d["node_flags"][rec] = 1 | (2 if node.terminal else 0)
        if node.terminal:
            return
        for a in node.child_actions:
            k = a._rec
            nrec = max(nrec, k + 1)
            d["parent"][k] = rec
            d["edge_n"][k] = a.n
            d["edge_W"][k] = float(np.asarray(a.W).reshape(-1)[0])
            d["edge_Q"][k] = float(np.asarray(a.Q).reshape(-1)[0])
            d["edge_action"][k] = np.float32(np.asarray(a.action).reshape(-1)[0])
            if hasattr(a, "child_node"):
                visit(a.child_node, k)

    visit(root, 0)
    d["node_r"][0] = 0.0   # a reused root keeps the reward of the edge that led to it; backprop never reads it
    d["n_records"] = np.int32(nrec)
    return d


def run_t1(case):
    """One T1 case: several independent trees, one reference MCTS object per tree."""
    cont = case["mode"] == 1
    in_dim = (2 if case["env_id"] == 4 else 3) if cont else {3: 2, 5: 6}.get(case["env_id"], 4)
    n_dist = 2 if cont else case["num_actions"]
    eng = O.OracleEngine(env_id=case["env_id"], mode=case["mode"], n_trees=1, n_sims=case["n_sims"], c_uct=case["c_uct"],
                         gamma=case["gamma"], epsilon=case["epsilon"], num_actions=case.get("num_actions", 0),
                         c_pw=case.get("c_pw", 1.0), kappa=case.get("kappa", 0.5), v_target=case["v_target"],
                         action_bound=case.get("action_bound", 2.0), seed=case["seed"])
    blob = O.make_weights(case["wseed"], in_dim, case["hidden"], n_dist, scale=case.get("wscale", 1.0))
    eng.set_weights(_capi.make_desc(in_dim, case["hidden"], n_dist, case["act"]), blob)
    R = eng.max_records
    out = {k: [] for k in ("counts", "Q", "actions", "v_target", "n_children", "root_state", "carry_in",
                           "n_records", "parent", "edge_n", "edge_W", "edge_Q", "edge_action", "node_n", "node_r", "node_V", "node_flags",
                           "child_n")}
    roots = np.asarray(case["roots"], dtype=np.float64)
    for ti, root in enumerate(roots):
        tree_id = case.get("tree_id_base", 0) + ti
        if cont and case["env_id"] == 4:
            env = MountainCarContinuousEnv(state=root)      # episodes end: terminal nodes in the continuous search
            root_obs = np.array(env.state)
        elif cont:
            env = PendulumEnv(state=root, version=1 if case["env_id"] == 2 else 0)
            root_obs = env._get_obs()
        elif case["env_id"] == 5:
            env = AcrobotEnv(state=root)                    # six observations, reward 0 on the terminal step
            root_obs = env._get_ob()
        else:
            env = MountainCarEnv(state=root) if case["env_id"] == 3 else CartPoleEnv(state=root)
            root_obs = np.array(env.state, dtype=np.float32)
        mcts = None
        carry = 0
        for step in range(case.get("reuse_steps", 1)):
            search_idx = case.get("search_idx", 0) + step
            model = OracleModel(eng, case["seed"], tree_id, search_idx, case.get("action_bound", 2.0))
            RM.random = EngineRandom(case["seed"], tree_id, search_idx)
            COUNTER["n"] = 0
            if mcts is None:
                if cont:
                    mcts = RM.MCTSContinuous(model=model, n_rollouts=case["n_sims"], c_uct=case["c_uct"], c_pw=case["c_pw"],
                                             kappa=case["kappa"], gamma=case["gamma"], epsilon=case["epsilon"],
                                             V_target_policy=case["v_target"], device="cpu", root_state=root_obs)
                else:
                    mcts = RM.MCTSDiscrete(model=model, num_actions=case["num_actions"], n_rollouts=case["n_sims"], c_uct=case["c_uct"],
                                           gamma=case["gamma"], epsilon=case["epsilon"], V_target_policy=case["v_target"],
                                           device="cpu", root_state=root_obs)
            else:
                mcts.model = model
            carry = 0 if mcts.root_node is None else mcts.root_node.n
            mcts.search(env)
            state, actions, counts, Q, V = mcts.return_results("max_visit")
            K = eng.kmax
            nc = len(counts)
            pad = lambda a, dt: np.concatenate([np.asarray(a, dtype=dt).reshape(-1), np.zeros(K - nc, dt)])  # noqa: E731
            out["counts"].append(pad(counts, np.int32))
            out["Q"].append(pad(np.array([np.asarray(q).reshape(-1)[0] for q in Q]), np.float64))
            out["actions"].append(pad(actions, np.float32))
            out["v_target"].append(np.float64(np.asarray(V).reshape(-1)[0]))
            out["n_children"].append(np.int32(nc))
            out["root_state"].append(np.asarray(env.azg_state(), dtype=np.float64))
            out["carry_in"].append(np.int32(carry))
            tree = dump_reference_tree(mcts.root_node, R)
            for k in ("n_records", "parent", "edge_n", "edge_W", "edge_Q", "edge_action", "node_n", "node_r", "node_V", "node_flags"):
                out[k].append(tree[k])
            cn = np.full(K, -1, np.int32)
            for a, act in enumerate(mcts.root_node.child_actions):
                if hasattr(act, "child_node"):
                    cn[a] = act.child_node.n
            out["child_n"].append(cn)
            if step + 1 < case.get("reuse_steps", 1):
                # the run loop of run_discrete.py:103-122: act -> Env.step -> mcts_forward
                a = int(np.argmax(counts))
                obs, r, done, _ = env.step(a)
                mcts.forward(a, obs)
                if done:
                    break
    res = {k: np.stack(v) for k, v in out.items()}
    res["case"] = np.array(repr(case))
    eng.close()
    return res


T1_CASES = {
    "t1_pendulum_v1_default": dict(env_id=2, mode=1, n_sims=200, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0.0,
                                   v_target="off_policy", hidden=[256, 256], act="elu", wseed=34, seed=34,
                                   roots=[[0.75, -0.5], [-2.9, 0.9], [3.0, 0.1]]),
    "t1_pendulum_v0_gamma": dict(env_id=1, mode=1, n_sims=60, c_uct=0.5, c_pw=0.8, kappa=0.6, gamma=0.99, epsilon=0.0,
                                 v_target="on_policy", hidden=[64], act="relu", wseed=7, seed=99, tree_id_base=1000,
                                 search_idx=5, roots=[[0.1, 0.0], [-1.5, -0.7]]),
    "t1_pendulum_v1_epsgreedy": dict(env_id=2, mode=1, n_sims=80, c_uct=0.1, c_pw=1, kappa=0.5, gamma=0.9, epsilon=0.25,
                                     v_target="greedy", hidden=[128, 128, 128], act="elu", wseed=11, seed=5,
                                     roots=[[2.0, 0.5], [-0.3, -0.9]]),
    "t1_cartpole_default": dict(env_id=0, mode=0, num_actions=2, n_sims=100, c_uct=1.5, gamma=1, epsilon=0.0,
                                v_target="off_policy", hidden=[128, 128], act="relu", wseed=34, seed=34,
                                roots=[[0.01, -0.02, 0.03, 0.04], [0.0, 0.0, 0.19, 0.8], [2.3, 1.0, 0.0, 0.0]]),
    "t1_cartpole_epsgreedy": dict(env_id=0, mode=0, num_actions=2, n_sims=40, c_uct=1.0, gamma=0.95, epsilon=0.1,
                                  v_target="on_policy", hidden=[64, 64], act="elu", wseed=3, seed=8, wscale=3.0,
                                  roots=[[0.02, 0.01, -0.15, -0.6], [-0.04, 0.03, 0.02, -0.01]]),
    "t1_cartpole_explore": dict(env_id=0, mode=0, num_actions=2, n_sims=120, c_uct=25.0, gamma=0.97, epsilon=0.0,
                                v_target="off_policy", hidden=[128, 128], act="relu", wseed=5, seed=1, wscale=2.0,
                                roots=[[0.01, -0.02, 0.03, 0.04], [0.5, 1.0, 0.15, 0.9], [-2.2, -1.5, -0.05, 0.2]]),
    "t1_cartpole_reuse": dict(env_id=0, mode=0, num_actions=2, n_sims=25, c_uct=1.5, gamma=1, epsilon=0.0,
                              v_target="off_policy", hidden=[128, 128], act="relu", wseed=34, seed=34, reuse_steps=4,
                              roots=[[0.03, 0.01, -0.02, 0.04], [-0.01, 0.02, 0.04, -0.03]]),
    # three actions (mcts.py:316-327 num_actions, 412-415): gym MountainCar-v0; roots in the valley, next to the flag (terminal
    # children), at the left wall and on the slope
    "t1_mountaincar_default": dict(env_id=3, mode=0, num_actions=3, n_sims=90, c_uct=0.8, gamma=0.99, epsilon=0.0,
                                   v_target="off_policy", hidden=[64, 64], act="relu", wseed=17, seed=21, wscale=2.0,
                                   roots=[[-0.5, 0.0], [0.43, 0.035], [-1.19, -0.03], [0.3, 0.05]]),
    # MCTSContinuous over an env whose episodes END (mcts.py:619-623 V = 0 for a terminal node, 682 `while not node.terminal`):
    # gym MountainCarContinuous-v0, action bound 1.  Roots one step from the flag (every child terminal: the whole search is traces
    # that end in existing terminal nodes), two and a few steps away (terminal nodes deeper in the tree), in the valley (none) and
    # running into the left wall; the second case discounts, explores with epsilon-greedy and another widening law (c_pw <= 1: with
    # c_pw > 1 the first trace widens the root again and leaves the root's initial action unvisited with a 0-d Q next to (1,)-shaped
    # ones, which the reference's own np.array(UCT) cannot stack under NumPy 2).
    "t1_mcc_terminal": dict(env_id=4, mode=1, n_sims=120, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0.0, v_target="off_policy",
                            hidden=[64, 64], act="elu", wseed=41, seed=43, wscale=2.0, action_bound=1.0,
                            roots=[[0.43, 0.03], [0.385, 0.04], [0.30, 0.055], [-0.5, 0.0], [-1.19, -0.04]]),
    "t1_mcc_terminal_eps": dict(env_id=4, mode=1, n_sims=90, c_uct=0.3, c_pw=0.9, kappa=0.65, gamma=0.97, epsilon=0.2,
                                v_target="on_policy", hidden=[128, 128], act="relu", wseed=42, seed=44, wscale=3.0, action_bound=1.0,
                                tree_id_base=300, search_idx=2, roots=[[0.40, 0.045], [0.34, 0.06], [0.25, 0.065]]),
    # gym Acrobot-v1: six observations (two k-steps in the network's first layer), three torques, Runge-Kutta dynamics, reward -1 per step
    # and 0 on the step that ends the episode; roots about to swing the tip over the line (terminal children), at rest, on the way up
    "t1_acrobot_default": dict(env_id=5, mode=0, num_actions=3, n_sims=70, c_uct=1.0, gamma=0.99, epsilon=0.0, v_target="off_policy",
                               hidden=[64, 64], act="relu", wseed=51, seed=53, wscale=2.0,
                               roots=[[1.9, 0.2, 2.0, 1.0], [0.05, -0.03, 0.02, 0.01], [1.373, -0.681, 2.48, 1.698], [-1.852, 0.74, -1.543, -1.538]]),
    "t1_acrobot_epsgreedy_reuse": dict(env_id=5, mode=0, num_actions=3, n_sims=40, c_uct=2.5, gamma=1, epsilon=0.15, v_target="on_policy",
                                       hidden=[128, 128], act="elu", wseed=52, seed=54, wscale=3.0, reuse_steps=3, tree_id_base=40,
                                       roots=[[1.373, -0.681, 2.48, 1.698], [-0.4, 0.6, 1.0, -1.0]]),
    "t1_mountaincar_epsgreedy_reuse": dict(env_id=3, mode=0, num_actions=3, n_sims=40, c_uct=2.0, gamma=1, epsilon=0.15,
                                           v_target="on_policy", hidden=[128, 128], act="elu", wseed=19, seed=22, wscale=3.0,
                                           reuse_steps=3, roots=[[-0.45, 0.01], [0.41, 0.04]]),
}


def set_policy_weights(pol, blob, in_dim, hidden, n_dist):
    """Load the flat blob (state_dict order) into a reference policy."""
    p = 0
    k = in_dim
    lin = [m for m in pol.trunk if isinstance(m, torch.nn.Linear)] + [pol.value_head, pol.dist_head]
    for m in lin:
        o, i = m.weight.shape
        m.weight.data = torch.from_numpy(blob[p:p + o * i].reshape(o, i).copy()); p += o * i
        m.bias.data = torch.from_numpy(blob[p:p + o].copy()); p += o
    assert p == blob.size


def set_layernorm_params(pol, seed):
    """Non-trivial LayerNorm weight/bias (torch initialises them to 1/0); returns them flattened per layer."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for m in pol.trunk:
        if isinstance(m, torch.nn.LayerNorm):
            w = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
            b = rng.uniform(-0.3, 0.3, m.bias.shape).astype(np.float32)
            m.weight.data = torch.from_numpy(w.copy()); m.bias.data = torch.from_numpy(b.copy())
            out.append((w, b))
    return out


def run_t2():
    out = {}
    rng = np.random.Generator(np.random.PCG64(2024))
    # continuous 2x256 elu squashed-normal (BASELINE config C) and the 3x128 elu default trunk
    for name, hidden, act in (("c256", [256, 256], "elu"), ("c128x3", [128, 128, 128], "elu"), ("c64relu", [64], "relu")):
        blob = O.make_weights(34, 3, hidden, 2)
        pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity=act,
                          num_components=1, action_bound=2.0)
        set_policy_weights(pol, blob, 3, hidden, 2)
        th = rng.uniform(-np.pi, np.pi, 64); thd = rng.uniform(-8, 8, 64)
        obs = np.stack([np.cos(th), np.sin(th), thd], 1).astype(np.float32)
        x = torch.from_numpy(obs)
        with torch.no_grad():
            pol.eval()
            mu, sigma, V = pol(x)
        out[f"{name}_obs"] = obs
        out[f"{name}_V"] = pol.predict_V(x).reshape(-1)
        out[f"{name}_mu"] = mu.numpy().reshape(-1)
        out[f"{name}_sigma"] = sigma.numpy().reshape(-1)
    # LayerNorm after every trunk activation (layernorm: true, policies.py:105-118); widths that are not multiples of 64
    for name, hidden, act in (("ln_c64", [64, 64], "elu"), ("ln_c100", [100, 60], "relu")):
        blob = O.make_weights(37, 3, hidden, 2, scale=2.0)
        pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity=act,
                          num_components=1, action_bound=2.0, layernorm=True)
        set_policy_weights(pol, blob, 3, hidden, 2)
        set_layernorm_params(pol, 38)
        th = rng.uniform(-np.pi, np.pi, 64); thd = rng.uniform(-8, 8, 64)
        obs = np.stack([np.cos(th), np.sin(th), thd], 1).astype(np.float32)
        with torch.no_grad():
            pol.eval()
            mu, sigma, V = pol(torch.from_numpy(obs))
        out[f"{name}_obs"] = obs
        out[f"{name}_V"] = V.numpy().reshape(-1)
        out[f"{name}_mu"] = mu.numpy().reshape(-1)
        out[f"{name}_sigma"] = sigma.numpy().reshape(-1)
    # the other trunk nonlinearities of alphazero/network/utils.py:5-14
    for act in ("leakyrelu", "relu6", "swish", "hardswish"):
        hidden = [64, 64]
        blob = O.make_weights(36, 3, hidden, 2, scale=3.0)
        pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity=act,
                          num_components=1, action_bound=2.0)
        set_policy_weights(pol, blob, 3, hidden, 2)
        th = rng.uniform(-np.pi, np.pi, 64); thd = rng.uniform(-8, 8, 64)
        obs = np.stack([np.cos(th), np.sin(th), thd], 1).astype(np.float32)
        with torch.no_grad():
            pol.eval()
            mu, sigma, V = pol(torch.from_numpy(obs))
        out[f"a_{act}_obs"] = obs
        out[f"a_{act}_V"] = V.numpy().reshape(-1)
        out[f"a_{act}_mu"] = mu.numpy().reshape(-1)
        out[f"a_{act}_sigma"] = sigma.numpy().reshape(-1)
    # the reference's default continuous head: 2-component Gaussian mixture (config/policy/ContinuousPolicy.yaml:7)
    for name, hidden, nc in (("g128x3", [128, 128, 128], 2), ("g64c3", [64, 64], 3)):
        blob = O.make_weights(35, 3, hidden, 3 * nc)
        pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                          num_components=nc, action_bound=2.0)
        set_policy_weights(pol, blob, 3, hidden, 3 * nc)
        th = rng.uniform(-np.pi, np.pi, 64); thd = rng.uniform(-8, 8, 64)
        obs = np.stack([np.cos(th), np.sin(th), thd], 1).astype(np.float32)
        with torch.no_grad():
            pol.eval()
            mu, sigma, log_coeff, V = pol(torch.from_numpy(obs))
        out[f"{name}_obs"] = obs
        out[f"{name}_V"] = V.numpy().reshape(-1)
        out[f"{name}_mu"] = mu.numpy()
        out[f"{name}_sigma"] = sigma.numpy()
        out[f"{name}_mix"] = torch.softmax(log_coeff, -1).numpy()
    for name, hidden, act in (("d128", [128, 128], "relu"), ("d64elu", [64, 64], "elu")):
        blob = O.make_weights(34, 4, hidden, 2)
        pol = make_policy(representation_dim=4, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity=act,
                          num_actions=2)
        set_policy_weights(pol, blob, 4, hidden, 2)
        obs = rng.uniform(-1, 1, (64, 4)).astype(np.float32) * np.array([2.4, 3.0, 0.21, 3.0], np.float32)
        x = torch.from_numpy(obs)
        out[f"{name}_obs"] = obs
        out[f"{name}_V"] = pol.predict_V(x).reshape(-1)
        out[f"{name}_pi"] = pol.predict_pi(x)
    return out


def run_t3():
    """Reference end to end with its real torch policies; torch.normal patched to the engine's noise stream."""
    out = {}
    seed = 34
    # continuous
    hidden = [256, 256]
    blob = O.make_weights(34, 3, hidden, 2)
    pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                      num_components=1, action_bound=2.0)
    set_policy_weights(pol, blob, 3, hidden, 2)
    roots = np.array([[0.75, -0.5], [-2.9, 0.9], [3.0, 0.1], [1.2, 0.3], [-0.4, -0.2], [2.2, 0.7]])
    counts_l, Q_l, act_l, V_l = [], [], [], []
    orig_normal = torch.normal
    for ti, root in enumerate(roots):
        state = {"n": 0}

        def fake_normal(mean, std, *a, **k):
            state["n"] += 1
            eps = O.normal(seed, ti, 0, state["n"])
            return mean + std * np.float32(eps)

        torch.normal = fake_normal
        COUNTER["n"] = 0
        env = PendulumEnv(state=root, version=1)
        m = RM.MCTSContinuous(model=pol, n_rollouts=100, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0, V_target_policy="off_policy",
                              device="cpu", root_state=env._get_obs())
        m.search(env)
        s, actions, counts, Q, V = m.return_results("max_visit")
        K = 10
        counts_l.append(np.asarray(counts, np.int32)[:K]); Q_l.append(np.array([np.asarray(q).reshape(-1)[0] for q in Q])[:K])
        act_l.append(np.asarray(actions, np.float32)[:K]); V_l.append(float(np.asarray(V).reshape(-1)[0]))
        assert len(counts) == K, len(counts)
    torch.normal = orig_normal
    out["c_roots"] = roots; out["c_counts"] = np.stack(counts_l); out["c_Q"] = np.stack(Q_l); out["c_actions"] = np.stack(act_l)
    out["c_v_target"] = np.array(V_l)
    # discrete
    hidden = [128, 128]
    blob = O.make_weights(34, 4, hidden, 2)
    pol = make_policy(representation_dim=4, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity="relu", num_actions=2)
    set_policy_weights(pol, blob, 4, hidden, 2)
    roots = np.array([[0.01, -0.02, 0.03, 0.04], [0.0, 0.0, 0.19, 0.8], [2.3, 1.0, 0.0, 0.0], [-0.03, 0.02, -0.01, 0.04],
                      [0.04, -0.04, 0.05, -0.05], [-1.0, -0.5, 0.1, 0.3]])
    counts_l, Q_l, V_l = [], [], []
    for ti, root in enumerate(roots):
        COUNTER["n"] = 0
        env = CartPoleEnv(state=root)
        m = RM.MCTSDiscrete(model=pol, num_actions=2, n_rollouts=100, c_uct=1.5, gamma=1, epsilon=0.0, V_target_policy="off_policy",
                            device="cpu", root_state=np.array(env.state, dtype=np.float32))
        m.search(env)
        s, actions, counts, Q, V = m.return_results("max_visit")
        counts_l.append(np.asarray(counts, np.int32)); Q_l.append(np.asarray(Q, np.float64)); V_l.append(float(V))
    out["d_roots"] = roots; out["d_counts"] = np.stack(counts_l); out["d_Q"] = np.stack(Q_l); out["d_v_target"] = np.array(V_l)
    # continuous, the reference's default head: 2-component Gaussian mixture (DiagonalGMMPolicy).  MixtureSameFamily.sample draws
    # the component with torch.multinomial and every component's Normal with one torch.normal call: both patched to the
    # engine's draws of the widening record (uniform = third Philox word, N(0,1) = the record's noise)
    hidden = [128, 128, 128]
    blob = O.make_weights(35, 3, hidden, 6)
    pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                      num_components=2, action_bound=2.0)
    set_policy_weights(pol, blob, 3, hidden, 6)
    roots = np.array([[0.75, -0.5], [-2.9, 0.9], [3.0, 0.1], [1.2, 0.3]])
    counts_l, Q_l, act_l, V_l, margin = [], [], [], [], []
    orig_multinomial = torch.multinomial
    for ti, root in enumerate(roots):
        state = {"n": 0}

        def fake_multinomial(probs, num_samples, replacement=False, **k):
            state["n"] += 1                                  # one widening record per sample_action call
            u = np.float32(O.gmm_u(seed, ti, 0, state["n"]))
            cum = np.cumsum(probs.detach().numpy().astype(np.float32), axis=-1, dtype=np.float32)
            margin.append(float(np.min(np.abs(cum[..., :-1] - u))))
            idx = (u >= cum[..., :-1]).sum(-1)
            return torch.as_tensor(idx, dtype=torch.long).reshape(probs.shape[:-1] + (1,))

        def fake_normal(mean, std, *a, **k):
            eps = O.normal(seed, ti, 0, state["n"])
            return mean + std * np.float32(eps)

        torch.multinomial = fake_multinomial
        torch.normal = fake_normal
        COUNTER["n"] = 0
        env = PendulumEnv(state=root, version=1)
        m = RM.MCTSContinuous(model=pol, n_rollouts=60, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0, V_target_policy="off_policy",
                              device="cpu", root_state=env._get_obs())
        m.search(env)
        s, actions, counts, Q, V = m.return_results("max_visit")
        K = 8
        assert len(counts) == K, len(counts)
        counts_l.append(np.asarray(counts, np.int32)); Q_l.append(np.array([np.asarray(q).reshape(-1)[0] for q in Q]))
        act_l.append(np.asarray(actions, np.float32).reshape(-1)); V_l.append(float(np.asarray(V).reshape(-1)[0]))
    torch.normal = orig_normal
    torch.multinomial = orig_multinomial
    assert min(margin) > 1e-4, min(margin)   # no component pick sits on a boundary of the cumulative mixture weights
    out["g_roots"] = roots; out["g_counts"] = np.stack(counts_l); out["g_Q"] = np.stack(Q_l); out["g_actions"] = np.stack(act_l)
    out["g_v_target"] = np.array(V_l)
    return out


def run_t2_wide():
    """T2 for the wide trunks: BASELINE config E's 4x1024 ELU network (policies.py:436-464: 1024-term dot products, torch's
    blocked sgemm against the engine's k-ordered fma chains) and a 2x512 one, 256 observations each."""
    out = {}
    rng = np.random.Generator(np.random.PCG64(2025))
    for name, hidden in (("c1024x4", [1024] * 4), ("c512x2", [512, 512])):
        blob = O.make_weights(34, 3, hidden, 2)
        pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                          num_components=1, action_bound=2.0)
        set_policy_weights(pol, blob, 3, hidden, 2)
        th = rng.uniform(-np.pi, np.pi, 256); thd = rng.uniform(-8, 8, 256)
        obs = np.stack([np.cos(th), np.sin(th), thd], 1).astype(np.float32)
        x = torch.from_numpy(obs)
        with torch.no_grad():
            pol.eval()
            mu, sigma, V = pol(x)
        out[f"{name}_obs"] = obs
        out[f"{name}_V"] = pol.predict_V(x).reshape(-1)
        out[f"{name}_mu"] = mu.numpy().reshape(-1)
        out[f"{name}_sigma"] = sigma.numpy().reshape(-1)
    return out


def _t3_pendulum_leg(hidden, roots, n_rollouts, K, seed=34):
    """The reference's MCTSContinuous.search (mcts.py:656-702) with its real torch policy on every root; torch.normal patched to
    the engine's noise of tree id = the root's index."""
    blob = O.make_weights(34, 3, hidden, 2)
    pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                      num_components=1, action_bound=2.0)
    set_policy_weights(pol, blob, 3, hidden, 2)
    counts_l, Q_l, act_l, V_l = [], [], [], []
    orig_normal = torch.normal
    try:
        for ti, root in enumerate(roots):
            state = {"n": 0}

            def fake_normal(mean, std, *a, **k):
                state["n"] += 1
                eps = O.normal(seed, ti, 0, state["n"])
                return mean + std * np.float32(eps)

            torch.normal = fake_normal
            COUNTER["n"] = 0
            env = PendulumEnv(state=root, version=1)
            m = RM.MCTSContinuous(model=pol, n_rollouts=n_rollouts, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0,
                                  V_target_policy="off_policy", device="cpu", root_state=env._get_obs())
            m.search(env)
            s, actions, counts, Q, V = m.return_results("max_visit")
            assert len(counts) == K, len(counts)
            counts_l.append(np.asarray(counts, np.int32)); Q_l.append(np.array([np.asarray(q).reshape(-1)[0] for q in Q]))
            act_l.append(np.asarray(actions, np.float32).reshape(-1)); V_l.append(float(np.asarray(V).reshape(-1)[0]))
    finally:
        torch.normal = orig_normal
    return np.stack(counts_l), np.stack(Q_l), np.stack(act_l), np.array(V_l)


def run_t3_full():
    """T3 at the headline search size: n_rollouts = 200 (BASELINE configs C and E).  Roots are the engine's own synthetic roots
    of global tree ids 0..15 (config C's first 16 trees, 2x256 ELU) and 0..3 (config E's, 4x1024 ELU) -- not hand-picked."""
    out = {}
    eng = O.OracleEngine(env_id=2, mode=1, n_trees=16, n_sims=200, c_uct=0.05, gamma=1.0, c_pw=1, kappa=0.5, seed=34)
    roots = eng.synthetic_roots()
    eng.close()
    c, q, a, v = _t3_pendulum_leg([256, 256], roots, 200, 15)
    out["c_roots"] = roots; out["c_counts"] = c; out["c_Q"] = q; out["c_actions"] = a; out["c_v_target"] = v
    c, q, a, v = _t3_pendulum_leg([1024] * 4, roots[:4], 200, 15)
    out["e_roots"] = roots[:4]; out["e_counts"] = c; out["e_Q"] = q; out["e_actions"] = a; out["e_v_target"] = v
    return out



# ---------------------------------------------------------------------------------------------------------------- T3 at scale
def _leaf_log():
    """Record, per trace, the engine record id of the node MCTS.backprop starts from (mcts.py:241-267)."""
    log = []
    orig = RM.MCTS.backprop

    def traced(node, gamma):
        rec = node.parent_action._rec if node.parent_action is not None else 0
        up = node.parent_action.parent_node.parent_action if node.parent_action is not None else None
        log.append(rec | ((up._rec if up is not None else 0) << 16))   # (leaf record, its parent node's record): pins the trace's path
        return orig(node, gamma)

    return log, orig, staticmethod(traced)


def _t3_scale_chunk(job):
    """One worker's share of a T3-at-scale leg: global tree ids [lo, hi) of `kind` ('c' / 'e': Pendulum-v1 continuous with the
    reference's DiagonalNormalPolicy; 'b': CartPole discrete with its DiscretePolicy)."""
    kind, hidden, n_rollouts, lo, hi, roots, seed = job
    torch.set_num_threads(1)
    cont = kind not in ("b", "d", "m", "a")
    car = kind == "m"           # gym MountainCar-v0: three actions, observation = (position, velocity)
    acro = kind == "a"          # gym Acrobot-v1: three actions, six observations
    mcc = kind == "h"           # gym MountainCarContinuous-v0: continuous search with terminal nodes (VERDICT r04 row h)
    gmm = kind == "g"           # the reference's default continuous policy: 2-component mixture (config/policy/ContinuousPolicy.yaml)
    eps = 0.1 if kind == "d" else 0.0   # the reference's default discrete search: epsilon-greedy 0.1 (config/mcts/MCTSDiscrete.yaml)
    if gmm:
        blob = O.make_weights(35, 3, hidden, 6)
        pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                          num_components=2, action_bound=2.0)
        set_policy_weights(pol, blob, 3, hidden, 6)
    elif mcc:
        blob = O.make_weights(34, 2, hidden, 2)
        pol = make_policy(representation_dim=2, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                          num_components=1, action_bound=1.0)
        set_policy_weights(pol, blob, 2, hidden, 2)
    elif cont:
        blob = O.make_weights(34, 3, hidden, 2)
        pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                          num_components=1, action_bound=2.0)
        set_policy_weights(pol, blob, 3, hidden, 2)
    elif car:
        blob = O.make_weights(34, 2, hidden, 3)
        pol = make_policy(representation_dim=2, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity="relu",
                          num_actions=3)
        set_policy_weights(pol, blob, 2, hidden, 3)
    elif acro:
        blob = O.make_weights(34, 6, hidden, 3)
        pol = make_policy(representation_dim=6, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity="relu",
                          num_actions=3)
        set_policy_weights(pol, blob, 6, hidden, 3)
    else:
        blob = O.make_weights(34, 4, hidden, 2)
        pol = make_policy(representation_dim=4, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity="relu",
                          num_actions=2)
        set_policy_weights(pol, blob, 4, hidden, 2)
    res = []
    orig_normal, orig_multinomial, orig_random = torch.normal, torch.multinomial, RM.random
    log, orig_bp, traced = _leaf_log()
    RM.MCTS.backprop = traced
    TIES["n"] = 0
    margins = []
    try:
        for ti in range(lo, hi):
            state = {"n": 0}

            def fake_normal(mean, std, *a, **k):
                if not gmm:
                    state["n"] += 1        # one widening record per sample_action call (mixture: counted by fake_multinomial)
                noise = O.normal(seed, ti, 0, state["n"])
                return mean + std * np.float32(noise)

            def fake_multinomial(probs, num_samples, replacement=False, **k):
                # MixtureSameFamily.sample (policies.py:656-668): component by inverse CDF with the engine's uniform of the record
                state["n"] += 1
                u = np.float32(O.gmm_u(seed, ti, 0, state["n"]))
                cum = np.cumsum(probs.detach().numpy().astype(np.float32), axis=-1, dtype=np.float32)
                margins.append(float(np.min(np.abs(cum[..., :-1] - u))))
                idx = (u >= cum[..., :-1]).sum(-1)
                return torch.as_tensor(idx, dtype=torch.long).reshape(probs.shape[:-1] + (1,))

            torch.normal = fake_normal
            if gmm:
                torch.multinomial = fake_multinomial
            if eps:
                RM.random = EngineRandom(seed, ti, 0)
            COUNTER["n"] = 0
            del log[:]
            if mcc:
                env = MountainCarContinuousEnv(state=roots[ti - lo])
                m = RM.MCTSContinuous(model=pol, n_rollouts=n_rollouts, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0,
                                      V_target_policy="off_policy", device="cpu", root_state=np.array(env.state))
            elif cont:
                env = PendulumEnv(state=roots[ti - lo], version=1)
                m = RM.MCTSContinuous(model=pol, n_rollouts=n_rollouts, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0,
                                      V_target_policy="off_policy", device="cpu", root_state=env._get_obs())
            elif car:
                env = MountainCarEnv(state=roots[ti - lo])
                m = RM.MCTSDiscrete(model=pol, num_actions=3, n_rollouts=n_rollouts, c_uct=0.8, gamma=0.99, epsilon=0.0,
                                    V_target_policy="off_policy", device="cpu", root_state=np.array(env.state, dtype=np.float32))
            elif acro:
                env = AcrobotEnv(state=roots[ti - lo])
                m = RM.MCTSDiscrete(model=pol, num_actions=3, n_rollouts=n_rollouts, c_uct=0.8, gamma=0.99, epsilon=0.0,
                                    V_target_policy="off_policy", device="cpu", root_state=env._get_ob())
            else:
                env = CartPoleEnv(state=roots[ti - lo])
                m = RM.MCTSDiscrete(model=pol, num_actions=2, n_rollouts=n_rollouts, c_uct=1.5, gamma=1, epsilon=eps,
                                    V_target_policy="off_policy", device="cpu", root_state=np.array(env.state, dtype=np.float32))
            m.search(env)
            s, actions, counts, Q, V = m.return_results("max_visit")
            res.append((np.asarray(counts, np.int32).reshape(-1), np.array([np.asarray(q).reshape(-1)[0] for q in Q], np.float64),
                        np.asarray(actions, np.float32).reshape(-1), float(np.asarray(V).reshape(-1)[0]), np.array(log, np.int32)))
    finally:
        torch.normal, torch.multinomial, RM.random = orig_normal, orig_multinomial, orig_random
        RM.MCTS.backprop = orig_bp
    assert not margins or min(margins) > 1e-6, min(margins)   # no component pick sits on a boundary of the cumulative mixture weights
    return lo, res, TIES["n"]


T3_SCALE = {   # tag: (env_id, mode, hidden, activation, n_rollouts, trees, engine kwargs)   -- BASELINE configs C, B, E
    "c": (2, 1, [256, 256], "elu", 200, 4096, dict(c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5)),
    "b": (0, 0, [128, 128], "relu", 100, 4096, dict(c_uct=1.5, gamma=1.0, num_actions=2)),
    "e": (2, 1, [1024] * 4, "elu", 200, 1024, dict(c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5)),
    # the reference's own DEFAULT configurations (config/mcts/*.yaml, config/policy/*.yaml): 2-component mixture head on a 3x128 ELU
    # trunk with 25 rollouts; 2x128 ReLU with 8 rollouts and epsilon-greedy 0.1 (draws: the engine's, injected as `random`)
    "g": (2, 1, [128, 128, 128], "elu", 25, 1024, dict(c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5)),
    "d": (0, 0, [128, 128], "relu", 8, 1024, dict(c_uct=1.5, gamma=1.0, num_actions=2, epsilon=0.1)),
    # three actions end to end: gym MountainCar-v0 with the reference's DiscretePolicy (2x64 ReLU), 60 rollouts, gamma 0.99
    "m": (3, 0, [64, 64], "relu", 60, 1024, dict(c_uct=0.8, gamma=0.99, num_actions=3)),
    # the continuous search over an env whose episodes end (mcts.py:619-623, 682): gym MountainCarContinuous-v0 with the reference's
    # DiagonalNormalPolicy (2x256 ELU, action bound 1), 120 rollouts, roots on the slope below the flag (mcc_scale_roots)
    "h": (4, 1, [256, 256], "elu", 120, 1024, dict(c_uct=0.05, gamma=1.0, c_pw=1.0, kappa=0.5, action_bound=1.0)),
    # six observations end to end: gym Acrobot-v1 with the reference's DiscretePolicy (2x64 ReLU), 50 rollouts, roots on the way up (acrobot_scale_roots)
    "a": (5, 0, [64, 64], "relu", 50, 1024, dict(c_uct=0.8, gamma=0.99, num_actions=3)),
}


def acrobot_scale_roots(synthetic):
    """Roots of the Acrobot leg: the engine's synthetic roots hang at rest (all four variables U(-0.1, 0.1)), hundreds of steps from the
    episode's end.  They are mapped onto the upswing: theta1 = 1.4 + 6 s0, theta2 = 8 s1, dtheta1 = 3 + 20 s2, dtheta2 = 20 s3 -- from
    a step to many steps below the line the tip has to cross (the same formula in tests/test_t3_scale.py)."""
    s = np.asarray(synthetic)
    return np.stack([1.4 + 6.0 * s[:, 0], 8.0 * s[:, 1], 3.0 + 20.0 * s[:, 2], 20.0 * s[:, 3]], 1)


def mcc_scale_roots(synthetic):
    """Roots of the MountainCarContinuous leg: the engine's synthetic roots start in the valley (position U(-0.6, -0.4), at rest), from
    where no 120-rollout tree ever reaches the flag.  They are mapped onto the slope below it: u = (position + 0.6) / 0.2 in [0, 1)
    -> position 0.25 + 0.199 u, velocity 0.02 + 0.05 frac(17 u): a step to a few steps away from the flag at 0.45 (the same formula in
    tests/test_t3_scale.py)."""
    u = (np.asarray(synthetic)[:, 0] + 0.6) / 0.2
    return np.stack([0.25 + 0.199 * u, 0.02 + 0.05 * ((17.0 * u) % 1.0)], 1)
T3_SCALE_FULL = 1024   # trees per leg whose Q / actions / value target are stored as well (visit counts: every tree)


def run_t3_scale(procs=8, only=None):
    """T3 at BASELINE scale (VERDICT r03 row g): the reference's MCTSContinuous.search / MCTSDiscrete.search (mcts.py:418-462,
    656-702) with its REAL torch policies (policies.py:340-352, 436-499) on the engine's own synthetic roots 0..N-1 of configs C
    (all 4096 trees, 2x256 ELU, 200 rollouts), B (all 4096 trees, CartPole 2x128 ReLU, 100 rollouts) and E (128 trees, 4x1024 ELU,
    200 rollouts).  Stored: visit counts of every tree (uint8); Q, actions and value target of the first 1024 trees of a leg.  The generator also runs the C oracle on the same roots;
    for every tree whose counts differ it stores the reference's per-trace leaf records, so that the test can locate the first
    diverging trace and show that the oracle's arg-max there was a near-tie."""
    import multiprocessing as mp
    out = {}
    for tag, (env_id, mode, hidden, act, n_roll, B, kw) in T3_SCALE.items():
        if only and tag not in only:
            continue
        eng = O.OracleEngine(env_id=env_id, mode=mode, n_trees=B, n_sims=n_roll, seed=34, **kw)
        roots = eng.synthetic_roots()
        if tag == "h":
            roots = mcc_scale_roots(roots)
        if tag == "a":
            roots = acrobot_scale_roots(roots)
            keep = np.array([not AcrobotEnv(state=r)._terminal() for r in roots])
            roots[~keep] = [1.0, 0.0, 0.5, 0.0]            # (a mapped root that is already above the line: replaced by a fixed one)
        n_dist = 6 if tag == "g" else (3 if tag in ("m", "a") else 2)
        in_dim = 6 if tag == "a" else (2 if tag in ("m", "h") else (4 if mode == 0 else 3))
        eng.set_weights(_capi.make_desc(in_dim, hidden, n_dist, act, num_components=2 if tag == "g" else 0),
                        O.make_weights(35 if tag == "g" else 34, in_dim, hidden, n_dist))
        eng.trace_enable()
        eng.search(roots)
        ro = eng.results()
        o_leaf, o_margin = eng.trace_get()
        eng.close()
        per = max(1, B // (procs * 4))
        jobs = [(tag, hidden, n_roll, lo, min(B, lo + per), roots[lo:min(B, lo + per)], 34) for lo in range(0, B, per)]
        with mp.get_context("fork").Pool(procs) as pool:
            parts = pool.map(_t3_scale_chunk, jobs)
        parts.sort(key=lambda x: x[0])
        ties = sum(p[2] for p in parts)
        assert ties == 0, f"t3 scale {tag}: {ties} exact arg-max ties in the reference run"
        rows = [r for p in parts for r in p[1]]
        K = ro["counts"].shape[1]
        F = min(B, T3_SCALE_FULL)
        counts = np.zeros((B, K), np.uint8); Q = np.zeros((F, K), np.float64); actions = np.zeros((F, K), np.float32)
        nch = np.zeros(B, np.uint8); vt = np.zeros(F, np.float64)
        mism, mism_leaf = [], []
        for i, (c, q, a, v, leaf) in enumerate(rows):
            k = len(c)
            assert c.max() < 256
            nch[i] = k; counts[i, :k] = c
            if i < F:
                Q[i, :k] = q; actions[i, :k] = a; vt[i] = v
            same = k == ro["n_children"][i] and (ro["counts"][i][:k] == c).all()
            if not same:
                mism.append(i); mism_leaf.append(leaf)
        out[f"{tag}_roots"] = roots; out[f"{tag}_counts"] = counts; out[f"{tag}_Q"] = Q; out[f"{tag}_n_children"] = nch
        out[f"{tag}_v_target"] = vt
        if mode == 1:
            out[f"{tag}_actions"] = actions
        out[f"{tag}_mismatch_ids"] = np.array(mism, np.int32)
        out[f"{tag}_mismatch_ref_leaf"] = np.stack(mism_leaf).astype(np.int32) if mism else np.zeros((0, n_roll), np.int32)
        finite = o_margin[np.isfinite(o_margin)]
        print(f"t3 scale {tag}: {B} trees, {len(mism)} with different visit counts (match rate {1 - len(mism) / B:.4f}); "
              f"oracle arg-max gaps: min {finite.min():.3e}, {int((finite < 1e-6).sum())} of {finite.size} below 1e-6", flush=True)
        for i, leaf in zip(mism, mism_leaf):
            d = int(np.argmax(leaf != o_leaf[i]))
            print(f"   tree {i}: first diverging trace {d}, oracle's tightest arg-max gap on it {o_margin[i, d]:.3e}")
    return out


def main_scale(only=None):
    """`scale`: every leg; `scale m g ...`: only the named legs, merged into the existing fixture."""
    path = os.path.join(HERE, "t3_scale.npz")
    res = {}
    if only and os.path.exists(path):
        with np.load(path) as z:
            res = {k: z[k] for k in z.files}
    res.update(run_t3_scale(only=only))
    np.savez_compressed(path, **res)
    print("t3_scale.npz", os.path.getsize(path), "bytes")


def main_wide():
    t5u = run_t5_update()
    np.savez_compressed(os.path.join(HERE, "t5_update.npz"), **t5u)
    print("t5 update", t5u["c_info"][-1].tolist(), t5u["da0c_info"][-1].tolist())
    t2w = run_t2_wide()
    np.savez_compressed(os.path.join(HERE, "t2_mlp_wide.npz"), **t2w)
    print("t2 wide keys", len(t2w))
    TIES["n"] = 0
    t3f = run_t3_full()
    assert TIES["n"] == 0, "t3 full: argmax tie occurred in the reference run"
    np.savez_compressed(os.path.join(HERE, "t3_full_rollouts.npz"), **t3f)
    print("t3 full c_counts[0]", t3f["c_counts"][0].tolist(), "e_counts[0]", t3f["e_counts"][0].tolist())


def run_t4():
    """Agent level: the reference's DiscreteAgent.act / ContinuousAgent.act (agents.py:257-303, 492-537) on top of the
    T1 machinery (oracle-MLP evaluator, engine noise), built without hydra by filling the attributes __init__ would set."""
    import random as pyrandom
    from alphazero.agent.agents import ContinuousAgent, DiscreteAgent
    out = {}
    seed = 34
    # continuous
    hidden = [256, 256]
    eng = O.OracleEngine(env_id=2, mode=1, n_trees=1, n_sims=25, c_uct=0.05, gamma=1.0, c_pw=1, kappa=0.5, seed=seed)
    eng.set_weights(_capi.make_desc(3, hidden, 2, "elu"), O.make_weights(34, 3, hidden, 2))
    ag = object.__new__(ContinuousAgent)
    ag.final_selection = "max_visit"; ag.epsilon = 0
    env = PendulumEnv(state=[1.0, 0.2], version=1)
    rows = []
    for t in range(3):
        model = OracleModel(eng, seed, 0, t, 2.0)
        RM.random = EngineRandom(seed, 0, t)
        COUNTER["n"] = 0
        ag.mcts = RM.MCTSContinuous(model=model, n_rollouts=25, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0,
                                    V_target_policy="off_policy", device="cpu", root_state=env._get_obs())
        rs = np.asarray(env.azg_state()).copy()
        action, s, actions, counts, Qs, V = ag.act(env)
        rows.append(dict(root=rs, action=action, state=s, actions=actions, counts=counts, Qs=Qs, V=np.asarray(V)))
        env.step(action)
    for k in rows[0]:
        out["c_" + k] = np.stack([r[k] for r in rows])
    out["c_dtypes"] = np.array(repr({k: (str(np.asarray(v).dtype), np.asarray(v).shape) for k, v in rows[0].items()}))
    # discrete, deterministic final action + tree reuse through mcts_forward
    hidden = [128, 128]
    eng = O.OracleEngine(env_id=0, mode=0, n_trees=1, n_sims=30, c_uct=25.0, gamma=0.97, num_actions=2, seed=seed)
    eng.set_weights(_capi.make_desc(4, hidden, 2, "relu"), O.make_weights(5, 4, hidden, 2, scale=2.0))
    ag = object.__new__(DiscreteAgent)
    ag.final_selection = "max_visits"; ag.temperature = 1.0
    env = CartPoleEnv(state=[0.01, -0.02, 0.03, 0.04])
    ag.mcts = RM.MCTSDiscrete(model=None, num_actions=2, n_rollouts=30, c_uct=25.0, gamma=0.97, epsilon=0.0,
                              V_target_policy="off_policy", device="cpu", root_state=np.array(env.state, dtype=np.float32))
    rows = []
    for t in range(4):
        ag.mcts.model = OracleModel(eng, seed, 0, t, 2.0)
        RM.random = EngineRandom(seed, 0, t)
        COUNTER["n"] = 0
        rs = np.asarray(env.azg_state()).copy()
        action, s, actions, counts, Qs, V = ag.act(env, deterministic=True)
        rows.append(dict(root=rs, action=np.asarray(action), state=s, actions=actions, counts=counts, Qs=Qs, V=np.asarray(V),
                         pi=np.asarray(__import__("alphazero.helpers", fromlist=["x"]).stable_normalizer(counts, 1.0))))
        obs, r, done, _ = env.step(int(action))
        ag.mcts_forward(int(action), obs)
    for k in rows[0]:
        out["d_" + k] = np.stack([r[k] for r in rows])
    out["d_dtypes"] = np.array(repr({k: (str(np.asarray(v).dtype), np.asarray(v).shape) for k, v in rows[0].items()}))
    return out


def run_t5():
    """Training side: get_train_data + the three losses of the reference on fixed batches (agents.py:319-392, 539-603)."""
    from alphazero.agent.losses import A0CLoss, A0CLossTuned, AlphaZeroLoss
    out = {}
    rng = np.random.Generator(np.random.PCG64(77))
    hidden = [64, 64]
    blob = O.make_weights(21, 3, hidden, 2)
    pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                      num_components=1, action_bound=2.0)
    set_policy_weights(pol, blob, 3, hidden, 2)
    B, K = 16, 5
    states = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
    actions = rng.uniform(-1.9, 1.9, (B, K)).astype(np.float32)
    counts = rng.integers(1, 9, (B, K)).astype(np.float32)
    V = rng.uniform(-1, 0, (B, 1)).astype(np.float32)
    lp, ent, vh = pol.get_train_data(torch.from_numpy(states), torch.from_numpy(actions))
    out.update(c_states=states, c_actions=actions, c_counts=counts, c_V=V, c_log_probs=lp.detach().numpy(), c_entropy=ent.detach().numpy(),
               c_V_hat=vh.detach().numpy())
    l = A0CLoss(tau=0.1, policy_coeff=0.1, alpha=0.5, value_coeff=1, reduction="mean")
    d = l(log_probs=lp, counts=torch.from_numpy(counts), entropy=ent, V=torch.from_numpy(V), V_hat=vh)
    out["c_a0c"] = np.array([float(d[k]) for k in ("loss", "policy_loss", "entropy_loss", "value_loss")])
    lt = A0CLossTuned(action_dim=1, alpha_init=1, lr=0.001, tau=0.1, policy_coeff=0.1, value_coeff=1, reduction="mean", grad_clip=0, device="cpu")
    d = lt(log_probs=lp, counts=torch.from_numpy(counts), entropy=ent, V=torch.from_numpy(V), V_hat=vh)
    out["c_a0c_tuned"] = np.array([float(d[k]) for k in ("loss", "policy_loss", "entropy_loss", "value_loss", "alpha_loss")] + [float(lt.alpha)])
    # mixture head
    blob = O.make_weights(23, 3, hidden, 6)
    pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                      num_components=2, action_bound=2.0)
    set_policy_weights(pol, blob, 3, hidden, 6)
    lp, ent, vh = pol.get_train_data(torch.from_numpy(out["c_states"]), torch.from_numpy(out["c_actions"]))
    out.update(g_log_probs=lp.detach().numpy(), g_entropy=ent.detach().numpy(), g_V_hat=vh.detach().numpy())
    # discrete
    blob = O.make_weights(22, 4, hidden, 2)
    pol = make_policy(representation_dim=4, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity="relu", num_actions=2)
    set_policy_weights(pol, blob, 4, hidden, 2)
    states = rng.uniform(-1, 1, (B, 4)).astype(np.float32)
    actions = np.tile(np.arange(2, dtype=np.float32), (B, 1))
    counts = rng.integers(0, 20, (B, 2)).astype(np.float32)
    V = rng.uniform(0, 10, (B, 1)).astype(np.float32)
    lp, ent, vh = pol.get_train_data(torch.from_numpy(states), torch.from_numpy(actions))
    logits, vh2 = pol._get_dist_params(torch.from_numpy(states))
    out.update(d_states=states, d_actions=actions, d_counts=counts, d_V=V, d_log_probs=lp.detach().numpy(), d_entropy=ent.detach().numpy(),
               d_V_hat=vh.detach().numpy())
    az = AlphaZeroLoss(policy_coeff=1.0, value_coeff=0.5, reduction="mean")
    d = az(logits, torch.softmax(torch.from_numpy(counts), dim=-1), vh2, torch.from_numpy(V))
    out["d_az"] = np.array([float(d[k]) for k in ("loss", "policy_loss", "value_loss")])
    return out


def run_t5_update():
    """The reference's own optimiser steps: ContinuousAgent.update (agents.py:539-603) with A0CLossTuned (losses.py:431-500) and
    DiscreteAgent.update (agents.py:319-392) with A0CLoss (the `counts += 1` branch), three consecutive
    steps each on fixed minibatches with the reference's RMSprop settings (config/optimizer/RMSProp.yaml) and gradient clipping.
    Agents are built without hydra by filling the attributes __init__ would set.  Stored: the minibatches, the loss dictionaries
    of every step, the first step's gradients and the parameters after the third step."""
    from alphazero.agent.agents import ContinuousAgent, DiscreteAgent
    from alphazero.agent.losses import A0CLoss, A0CLossTuned
    out = {}
    rng = np.random.Generator(np.random.PCG64(78))
    hidden = [64, 64]
    B, K, STEPS = 16, 5, 3

    def flat(params, grads=False):
        return np.concatenate([(q.grad if grads else q.data).detach().numpy().ravel() for q in params])

    def drive(tag, ag, batches, keys):
        infos, g0 = [], None
        for i, b in enumerate(batches):
            info = ag.update(tuple(np.copy(x) for x in b))
            infos.append([info[k] for k in keys])
            if i == 0:
                g0 = flat(list(ag.nn.parameters()), grads=True)
        out[f"{tag}_info"] = np.array(infos, np.float64)
        out[f"{tag}_grad0"] = g0
        out[f"{tag}_params"] = flat(list(ag.nn.parameters()))

    def opt(pol):
        return torch.optim.RMSprop(pol.parameters(), lr=0.001, momentum=0, weight_decay=0, alpha=0.9, eps=1e-10)

    # continuous, A0CLossTuned
    pol = make_policy(representation_dim=3, action_dim=1, distribution="normal", hidden_dimensions=hidden, nonlinearity="elu",
                      num_components=1, action_bound=2.0)
    set_policy_weights(pol, O.make_weights(21, 3, hidden, 2), 3, hidden, 2)
    batches = [(rng.uniform(-1, 1, (B, 3)).astype(np.float32), rng.uniform(-1.9, 1.9, (B, K)).astype(np.float32),
                rng.integers(1, 9, (B, K)).astype(np.float32), rng.uniform(-1, 0, (B, K, 1)), rng.uniform(-1, 0, B)) for _ in range(STEPS)]
    ag = object.__new__(ContinuousAgent)
    ag.nn = pol; ag.device = torch.device("cpu"); ag.clip = 0.5; ag.optimizer = opt(pol)
    ag.loss = A0CLossTuned(action_dim=1, alpha_init=1, lr=0.001, tau=0.1, policy_coeff=0.1, value_coeff=1, reduction="mean", grad_clip=0.5, device="cpu")
    drive("c", ag, batches, ("loss", "policy_loss", "entropy_loss", "value_loss", "alpha_loss"))
    out["c_alpha"] = np.array(float(ag.loss.alpha))
    for i, name in enumerate(("states", "actions", "counts", "Qs", "V")):
        out[f"c_{name}"] = np.stack([b[i] for b in batches])
    # discrete, AlphaZeroLoss and A0CLoss
    dbatches = [(rng.uniform(-1, 1, (B, 4)).astype(np.float32), np.tile(np.arange(2, dtype=np.float32), (B, 1)),
                 rng.integers(0, 20, (B, 2)).astype(np.float32), rng.uniform(0, 10, (B, 2)), rng.uniform(0, 10, B)) for _ in range(STEPS)]
    for i, name in enumerate(("states", "actions", "counts", "Qs", "V")):
        out[f"d_{name}"] = np.stack([b[i] for b in dbatches])
    # (DiscreteAgent.update with AlphaZeroLoss cannot be captured: agents.py:380-381 hands the Categorical that DiscretePolicy.forward
    #  returns to F.cross_entropy, which raises TypeError in the reference itself; that loss is pinned on logits by run_t5)
    for tag, loss, keys in (("da0c", A0CLoss(tau=0.1, policy_coeff=1, alpha=1, value_coeff=1, reduction="mean"),
                             ("loss", "policy_loss", "entropy_loss", "value_loss")),):
        pol = make_policy(representation_dim=4, action_dim=1, distribution="discrete", hidden_dimensions=hidden, nonlinearity="relu", num_actions=2)
        set_policy_weights(pol, O.make_weights(22, 4, hidden, 2), 4, hidden, 2)
        ag = object.__new__(DiscreteAgent)
        ag.nn = pol; ag.device = torch.device("cpu"); ag.clip = 0; ag.optimizer = opt(pol); ag.loss = loss
        drive(tag, ag, dbatches, keys)
    return out


def run_t6():
    """The reference's ReplayBuffer (buffers.py:40-127) driven through wrap-around and two epochs of minibatches under a
    fixed numpy seed: slot contents after every store and the sample ids of every batch."""
    from alphazero.agent.buffers import ReplayBuffer
    out = {}
    buf = ReplayBuffer(max_size=7, batch_size=3)
    slots = []
    for i in range(17):
        buf.store((np.full(3, i, np.float32), np.full(2, i, np.float32), np.full(2, i, np.float32), np.full(2, i, np.float32), np.float64(i)))
        row = [int(e[4]) for e in buf.experience] + [-1] * (7 - len(buf.experience))
        slots.append(row + [buf.insert_index, buf.size])
    out["slots"] = np.array(slots)
    np.random.seed(123)
    buf.reshuffle()
    batches = []
    for epoch in range(2):
        for b in buf:
            ids = np.asarray(b[4]).reshape(-1).astype(np.int64)
            batches.append(np.concatenate([[epoch, len(ids)], ids, [-1] * (8 - len(ids))]))
            assert tuple(np.asarray(b[0]).shape) == (len(ids), 3)
    out["batches"] = np.array(batches)
    return out


T7_CASES = {
    # Pendulum, the default rule: most visited root action; episodes end by length (Pendulum never terminates)
    "pendulum": dict(env_id=2, mode=1, n_sims=20, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0.0, v_target="off_policy",
                     hidden=[64, 64], act="elu", wseed=3, seed=11, tree_id_base=5, n_games=3, n_steps=9, max_len=4),
    # final_selection max_value + the agent's epsilon-greedy, Pendulum-v0, discounting, eps-greedy inside the search too
    "pendulum_maxvalue_eps": dict(env_id=1, mode=1, n_sims=24, c_uct=0.1, c_pw=1, kappa=0.6, gamma=0.97, epsilon=0.1,
                                  v_target="on_policy", hidden=[64], act="relu", wseed=4, seed=12, tree_id_base=0, n_games=3,
                                  n_steps=8, max_len=5, final_selection="max_value", agent_eps=0.4),
    # CartPole: deterministic final action, tree reuse through mcts_forward, episodes cut by length
    "cartpole_det": dict(env_id=0, mode=0, num_actions=2, n_sims=16, c_uct=20.0, gamma=0.97, epsilon=0.0, v_target="off_policy",
                         hidden=[64, 64], act="relu", wseed=3, wscale=2.0, seed=13, tree_id_base=2, n_games=3, n_steps=14, max_len=6,
                         det=True),
    # sampled final action (temperature 1), long enough for real terminations
    "cartpole_sampled": dict(env_id=0, mode=0, num_actions=2, n_sims=12, c_uct=30.0, gamma=0.9, epsilon=0.0, v_target="on_policy",
                             hidden=[64, 64], act="relu", wseed=6, wscale=3.0, seed=14, tree_id_base=0, n_games=4, n_steps=60,
                             max_len=200, det=False),
    # sampled with temperature 0.5
    "cartpole_temp": dict(env_id=0, mode=0, num_actions=2, n_sims=16, c_uct=5.0, gamma=0.97, epsilon=0.0, v_target="off_policy",
                          hidden=[64], act="relu", wseed=7, wscale=2.0, seed=15, tree_id_base=1, n_games=3, n_steps=12, max_len=5,
                          det=False, temperature=0.5),
    # final_selection max_value, sampled from the normalised Qs (temperature 1)
    "cartpole_maxvalue": dict(env_id=0, mode=0, num_actions=2, n_sims=16, c_uct=5.0, gamma=0.97, epsilon=0.0, v_target="off_policy",
                              hidden=[64], act="relu", wseed=8, wscale=2.0, seed=16, tree_id_base=0, n_games=3, n_steps=10, max_len=7,
                              det=False, final_selection="max_value"),
    # three actions (gym MountainCar-v0): sampled final action, tree reuse through mcts_forward, episodes cut by length
    # MountainCarContinuous: the first episode of a game starts below the flag (first_roots, uploaded after selfplay_begin), so
    # that episodes END by reaching it (+100) and by length; searches meet terminal nodes; later episodes start in the valley
    "mcc": dict(env_id=4, mode=1, n_sims=24, c_uct=0.05, c_pw=1, kappa=0.5, gamma=1, epsilon=0.0, v_target="off_policy",
                hidden=[64, 64], act="elu", wseed=10, wscale=2.0, seed=18, tree_id_base=7, n_games=4, n_steps=10, max_len=6,
                action_bound=1.0, first_roots=[[0.41, 0.04], [0.33, 0.06], [0.36, 0.02], [-0.5, 0.0]]),
    # Acrobot-v1: games start on the upswing (first_roots), so that episodes END by the tip crossing the line (reward 0 on that step)
    "acrobot_sampled": dict(env_id=5, mode=0, num_actions=3, n_sims=16, c_uct=2.0, gamma=0.98, epsilon=0.0, v_target="off_policy",
                            hidden=[64, 64], act="relu", wseed=11, wscale=3.0, seed=19, tree_id_base=4, n_games=3, n_steps=12, max_len=6,
                            det=False, first_roots=[[1.9, 0.2, 2.0, 1.0], [1.373, -0.681, 2.48, 1.698], [0.05, -0.03, 0.02, 0.01]]),
    "mountaincar_sampled": dict(env_id=3, mode=0, num_actions=3, n_sims=18, c_uct=2.0, gamma=0.98, epsilon=0.0, v_target="off_policy",
                                hidden=[64, 64], act="relu", wseed=9, wscale=3.0, seed=17, tree_id_base=3, n_games=3, n_steps=16, max_len=7,
                                det=False),
}


def run_t7(case):
    """The reference's run loop for every game of a self-play batch (see the module docstring).  What the engine defines and
    the reference receives through patched RNG entry points: the reset state of an episode (stands in for Env.reset()), the
    search noise / eps-greedy draws (as in T1) and the final-action draw of a step (np.random.choice / random.random in
    alphazero/agent/agents.py, patched to numpy's own inverse-CDF rule on the engine's uniform)."""
    import alphazero.agent.agents as RA
    from alphazero.agent.buffers import ReplayBuffer
    cont = case["mode"] == 1
    mcc = case["env_id"] == 4
    bound = case.get("action_bound", 2.0)
    in_dim = (2 if mcc else 3) if cont else {3: 2, 5: 6}.get(case["env_id"], 4)
    acro = case["env_id"] == 5
    n_dist = 2 if cont else case["num_actions"]
    seed = case["seed"]
    eng = O.OracleEngine(env_id=case["env_id"], mode=case["mode"], n_trees=1, n_sims=case["n_sims"], c_uct=case["c_uct"],
                         gamma=case["gamma"], epsilon=case["epsilon"], num_actions=case.get("num_actions", 0),
                         c_pw=case.get("c_pw", 1.0), kappa=case.get("kappa", 0.5), v_target=case["v_target"], seed=seed,
                         action_bound=bound)
    eng.set_weights(_capi.make_desc(in_dim, case["hidden"], n_dist, case["act"]),
                    O.make_weights(case["wseed"], in_dim, case["hidden"], n_dist, scale=case.get("wscale", 1.0)))
    K, So = eng.kmax, eng.s_obs
    fs = case.get("final_selection", "max_visit")
    n_steps, max_len = case["n_steps"], case["max_len"]
    G = case["n_games"]
    rows = np.zeros((n_steps, G, So + 3 * K + 1), np.float64)
    act_idx = np.zeros((n_steps, G), np.int32)
    act_val = np.zeros((n_steps, G), np.float64)
    root_before = np.zeros((n_steps, G, eng.s_env), np.float64)
    carry_in = np.zeros((n_steps, G), np.int32)
    fsum, fcnt = np.zeros(G), np.zeros(G, np.int32)
    final_state = np.zeros((G, eng.s_env))
    step_box = {"gt": 0, "step": 0}

    class ActRandom:   # `random` inside alphazero.agent.agents (epsilon_greedy, agents.py:487)
        @staticmethod
        def random():
            return O.act_draw(seed, step_box["gt"], step_box["step"])[0]

    def fake_choice(a, p=None, **kw):   # np.random.choice inside alphazero.agent.agents (agents.py:299, 301, 488)
        u01, u, word = O.act_draw(seed, step_box["gt"], step_box["step"])
        if p is None:
            a = np.asarray(a)
            return a[word % len(a)]
        cdf = np.asarray(p, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        return int(cdf.searchsorted(u, side="right"))

    orig_choice, orig_random = np.random.choice, RA.random
    np.random.choice = fake_choice
    RA.random = ActRandom
    try:
        for g in range(G):
            gt = case["tree_id_base"] + g
            step_box["gt"] = gt
            if cont:
                env = MountainCarContinuousEnv() if mcc else PendulumEnv(version=1 if case["env_id"] == 2 else 0)
                ag = object.__new__(RA.ContinuousAgent)
                ag.final_selection = fs; ag.epsilon = case.get("agent_eps", 0)
                ag.mcts = RM.MCTSContinuous(model=None, n_rollouts=case["n_sims"], c_uct=case["c_uct"], c_pw=case["c_pw"],
                                            kappa=case["kappa"], gamma=case["gamma"], epsilon=case["epsilon"],
                                            V_target_policy=case["v_target"], device="cpu", root_state=None)
            else:
                env = AcrobotEnv() if acro else (MountainCarEnv() if case["env_id"] == 3 else CartPoleEnv())
                ag = object.__new__(RA.DiscreteAgent)
                ag.final_selection = fs; ag.temperature = case.get("temperature", 1.0)
                ag.mcts = RM.MCTSDiscrete(model=None, num_actions=case["num_actions"], n_rollouts=case["n_sims"], c_uct=case["c_uct"],
                                          gamma=case["gamma"], epsilon=case["epsilon"], V_target_policy=case["v_target"],
                                          device="cpu", root_state=None)
            buffer = ReplayBuffer(max_size=10 ** 6, batch_size=8)
            step, episode = 0, 0
            while step < n_steps:
                # Env.reset() (run_*.py: `state = Env.reset()`), with the engine's reset state of (game, episode)
                rs = O.reset_state(seed, gt, episode, not cont, env_id=case["env_id"])
                if episode == 0 and "first_roots" in case:
                    rs = np.asarray(case["first_roots"][g], np.float64)   # (the test uploads these roots after selfplay_begin)
                env.state = np.asarray(rs, np.float64) if cont else tuple(float(v) for v in rs)
                state = (np.array(env.state) if mcc else env._get_obs()) if cont else (env._get_ob() if acro else np.array(env.state, dtype=np.float32))
                R = 0.0
                ag.reset_mcts(root_state=state)
                for t in range(max_len):
                    step_box["step"] = step
                    ag.mcts.model = OracleModel(eng, seed, gt, step, bound)
                    RM.random = EngineRandom(seed, gt, step)
                    COUNTER["n"] = 0
                    root_before[step, g] = env.azg_state()
                    carry_in[step, g] = 0 if ag.mcts.root_node is None else ag.mcts.root_node.n
                    if cont:
                        action, s, actions, counts, Qs, V = ag.act(Env=env)
                    else:
                        action, s, actions, counts, Qs, V = ag.act(Env=env, deterministic=case["det"])
                    buffer.store((s, actions, counts, Qs, V))
                    nc = len(counts)
                    row = rows[step, g]
                    row[:So] = np.asarray(s, np.float64).reshape(-1)
                    row[So:So + nc] = np.asarray(actions, np.float64).reshape(-1)
                    row[So + K:So + K + nc] = counts
                    row[So + 2 * K:So + 2 * K + nc] = np.asarray(Qs, np.float64).reshape(-1)
                    row[So + 3 * K] = float(np.asarray(V).reshape(-1)[0])
                    if cont:
                        act_val[step, g] = float(np.asarray(action).reshape(-1)[0])
                        act_idx[step, g] = int(np.where(np.atleast_1d(actions) == action[0])[0][0])
                    else:
                        act_idx[step, g] = int(action)
                        act_val[step, g] = float(action)
                    state, step_reward, terminal, _ = env.step(action)
                    R += float(np.asarray(step_reward).reshape(-1)[0])
                    step += 1
                    if terminal or t == max_len - 1:
                        fsum[g] += R
                        fcnt[g] += 1
                        break
                    if cont:
                        ag.reset_mcts(root_state=state)       # run_continuous.py:140-142
                    else:
                        ag.mcts_forward(action, state)        # run_discrete.py:121-122
                    if step >= n_steps:
                        break
                else:
                    pass
                if step >= n_steps and not (terminal or t == max_len - 1):
                    final_state[g] = env.azg_state()          # stopped in the middle of an episode
                    break
                episode += 1
                if step >= n_steps:
                    # the device resets the game in the same step that ends the episode
                    final_state[g] = O.reset_state(seed, gt, episode, not cont, env_id=case["env_id"])
            assert len(buffer) == n_steps
            # the buffer holds what the rows hold (buffer.store is the reference's own)
            for i, exp in enumerate(buffer.experience):
                np.testing.assert_array_equal(np.asarray(exp[2]).reshape(-1), rows[i, g, So + K:So + K + len(exp[2])])
    finally:
        np.random.choice = orig_choice
        RA.random = orig_random
    eng.close()
    return dict(rows=rows, act_idx=act_idx, act_val=act_val, root_before=root_before, carry_in=carry_in, fsum=fsum, fcnt=fcnt,
                final_state=final_state, case=np.array(repr(case)))


def main_t7(only=None):
    for name, case in T7_CASES.items():
        if only and name not in only:
            continue
        TIES["n"] = 0
        t7 = run_t7(case)
        assert TIES["n"] == 0, f"t7 {name}: argmax tie occurred in the reference run; pick other inputs"
        np.savez_compressed(os.path.join(HERE, f"t7_selfplay_{name}.npz"), **t7)
        print("t7", name, "episodes", t7["fcnt"].tolist(), "returns", np.round(t7["fsum"], 3).tolist(), "actions[:, 0]", t7["act_idx"][:, 0].tolist())


def main():
    if sys.argv[1:2] == ["t7"]:   # only the self-play tier, or only its named cases (the other fixtures are left untouched)
        return main_t7(sys.argv[2:])
    if sys.argv[1:2] == ["t1"]:   # only the named T1 cases: python gen_golden.py t1 t1_mountaincar_default ...
        for name in sys.argv[2:]:
            TIES["n"] = 0
            res = run_t1(T1_CASES[name])
            assert TIES["n"] == 0, f"{name}: argmax tie occurred in the reference run; pick other inputs"
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
            print(name, "records", res["n_records"].tolist(), "counts", res["counts"].tolist())
        return
    if sys.argv[1:2] == ["scale"] and len(sys.argv) > 2:   # only the named legs of t3_scale.npz
        return main_scale(sys.argv[2:])
    if sys.argv[1:] == ["scale"]:  # only t3_scale.npz: the reference with its torch policies on 4096 + 4096 + 128 synthetic roots
        return main_scale()
    if sys.argv[1:] == ["wide"]:   # only the wide-network T2 cases and the n_rollouts = 200 T3 legs
        return main_wide()
    for name, case in T1_CASES.items():
        TIES["n"] = 0
        res = run_t1(case)
        assert TIES["n"] == 0, f"{name}: argmax tie occurred in the reference run; pick other inputs"
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **res)
        print(name, "records", res["n_records"].tolist(), "counts[0]", res["counts"][0].tolist(), os.path.getsize(path), "bytes")
    t2 = run_t2()
    np.savez_compressed(os.path.join(HERE, "t2_mlp_torch.npz"), **t2)
    print("t2 keys", len(t2))
    TIES["n"] = 0
    t3 = run_t3()
    np.savez_compressed(os.path.join(HERE, "t3_end_to_end.npz"), **t3)
    t4 = run_t4()
    np.savez_compressed(os.path.join(HERE, "t4_agent_act.npz"), **t4)
    print("t4", t4["c_dtypes"], t4["d_dtypes"], t4["d_action"].tolist(), t4["d_counts"].tolist())
    t5 = run_t5()
    np.savez_compressed(os.path.join(HERE, "t5_training.npz"), **t5)
    print("t5", t5["c_a0c"], t5["c_a0c_tuned"], t5["d_az"])
    t6 = run_t6()
    np.savez_compressed(os.path.join(HERE, "t6_buffer.npz"), **t6)
    print("t6", t6["slots"][-1].tolist(), t6["batches"].tolist())
    print("t3 ties", TIES["n"], "c_counts[0]", t3["c_counts"][0].tolist(), "d_counts", t3["d_counts"].tolist())
    main_t7()
    main_wide()
    main_scale()

if __name__ == "__main__":
    main()
