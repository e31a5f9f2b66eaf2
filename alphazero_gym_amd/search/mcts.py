"""MCTS with the reference's class surface (alphazero/search/mcts.py), executed by the MI355X engine.

``MCTSDiscrete`` / ``MCTSContinuous`` keep the reference's constructor kwargs (mcts.py:316-327, 537-549), the
attributes the agents poke (``root_node``, ``root_state``, ``n_rollouts``, ``c_uct``, ``gamma``), ``search(Env)``,
``return_results(final_selection)`` and (discrete) ``forward(action, state)``; ``model`` is the torch policy whose
weights the engine evaluates on the GPU.  One ``search`` call is ONE kernel launch that runs all ``n_rollouts`` traces.

``BatchedMCTS`` is the native interface: B independent trees per call (``Env`` may also be a list of environments
for the two classes above).  There is no CPU fallback; constructing an engine without the HIP library or a GPU raises.
"""
import warnings
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from .. import _capi

PENDULUM_R_SCALE = _capi.PENDULUM_R_SCALE  # scales the step reward between -1 and 0 (mcts.py:19-20)


def env_signature(env) -> Tuple[int, np.ndarray]:
    """(engine env id, float64 internal state) of a supported environment object."""
    u = getattr(env, "unwrapped", env)
    if hasattr(u, "azg_env_id"):
        return int(u.azg_env_id), np.asarray(u.azg_state(), dtype=np.float64)
    name = type(u).__name__
    if name == "CartPoleEnv":
        return _capi.ENV_CARTPOLE, np.asarray(u.state, dtype=np.float64)
    if name == "MountainCarEnv":
        return _capi.ENV_MOUNTAINCAR, np.asarray(u.state, dtype=np.float64)
    if name == "AcrobotEnv":
        return _capi.ENV_ACROBOT, np.asarray(u.state, dtype=np.float64)
    if name == "Continuous_MountainCarEnv":
        return _capi.ENV_MOUNTAINCAR_CONT, np.asarray(u.state, dtype=np.float64)
    if name == "PendulumEnv":
        spec = getattr(getattr(u, "spec", None), "id", "") or ""
        return (_capi.ENV_PENDULUM_V0 if spec.endswith("v0") else _capi.ENV_PENDULUM_V1), np.asarray(u.state, dtype=np.float64)
    raise NotImplementedError(
        f"{name}: the engine steps CartPole, MountainCar, MountainCarContinuous, Acrobot and Pendulum in closed form on the GPU; other "
        "environments are not supported")


def env_observation(env_id: int, st: np.ndarray) -> np.ndarray:
    """float32 observation the reference's nodes keep as ``state`` for an engine state: Acrobot observes (cos, sin) of both joint
    angles and the two velocities (gym acrobot.py _get_ob), Pendulum (cos, sin, thdot); the others observe their state."""
    st = np.asarray(st, dtype=np.float64)
    if env_id == _capi.ENV_ACROBOT:
        return np.array([np.cos(st[0]), np.sin(st[0]), np.cos(st[1]), np.sin(st[1]), st[2], st[3]], dtype=np.float32)
    if env_id in (_capi.ENV_PENDULUM_V0, _capi.ENV_PENDULUM_V1):
        return np.array([np.cos(st[0]), np.sin(st[0]), st[1]], dtype=np.float32)
    return st.astype(np.float32)


def _weights_version(model) -> Tuple:
    """Changes whenever a parameter is written in place (optimiser step, copy_, broadcast: ``_version``) or rebound to
    new storage (``p.data = ...``, ``load_state_dict(assign=True)``: ``data_ptr``)."""
    return tuple((id(p), p._version, p.data_ptr()) for p in model.parameters())


def device_ordinal(device) -> int:
    """GPU ordinal for the reference's ``device`` kwarg (mcts.py:316-327).  "cuda:N" -> N; "cuda" -> torch's current device;
    anything else (the reference's configs say "cpu": that is where ITS network ran) -> LOCAL_RANK, so that one agent per
    rank lands on its own GPU, else 0."""
    import os
    try:
        import torch
        d = torch.device(device) if device is not None else None
        if d is not None and d.type == "cuda":
            if d.index is not None:
                return int(d.index)
            if torch.cuda.is_available():
                return int(torch.cuda.current_device())
    except (RuntimeError, TypeError):
        pass
    return int(os.environ.get("LOCAL_RANK", "0"))


class BatchedMCTS:
    """B independent trees searched in one launch.  kwargs = the reference's MCTS kwargs + batch geometry."""

    def __init__(self, model, *, env_id: int, mode: int, n_trees: int, n_rollouts: int, c_uct: float, gamma: float,
                 epsilon: float = 0.0, num_actions: int = 0, c_pw: float = 1.0, kappa: float = 0.5,
                 V_target_policy: str = "off_policy", action_bound: float = 2.0, seed: int = 34, tree_id_base: int = 0,
                 device_id: int = 0):
        from .. import _native   # raises if libazgym_hip.so is missing

        self.model = model
        self.engine = _native.HipEngine(env_id=env_id, mode=mode, n_trees=n_trees, n_sims=n_rollouts, c_uct=c_uct, gamma=gamma,
                                        epsilon=epsilon, num_actions=num_actions, c_pw=c_pw, kappa=kappa, v_target=V_target_policy,
                                        action_bound=action_bound, seed=seed, tree_id_base=tree_id_base, device_id=device_id)
        self._version = None
        self._cliff_warned = False
        self.sync_weights()

    def sync_weights(self, force: bool = False) -> None:
        """Push the model's weights to the engine when they changed (after every optimiser step)."""
        v = _weights_version(self.model)
        if force or v != self._version:
            self.engine.set_policy(self.model)
            self._version = v

    def search(self, root_states: np.ndarray, root_n_carry: Optional[np.ndarray] = None) -> None:
        self.sync_weights()
        self.engine.search(root_states, root_n_carry)
        if not self._cliff_warned and hasattr(self.engine, "search_info"):   # (the tests' CPU double of the engine has no kernel forms)
            info = self.engine.search_info()
            if info["kernel_form"] == "persistent" and info["tree_storage"] == "global" and info["lds_exit"] != "forced":
                self._cliff_warned = True
                warnings.warn(f"the trees of this search do not fit LDS residency ({info['lds_exit']}: {info['max_records']} records per tree, up to "
                              f"{info['max_children']} children per node): they are kept in global memory -- same results, slower tree walk "
                              "(BatchedMCTS.last_search_info)", RuntimeWarning, stacklevel=2)

    @property
    def last_search_info(self) -> dict:
        """What the last search ran as (azg_search_info): ``kernel_form`` "persistent" / "per_layer" / "team", ``tree_storage`` "lds8" /
        "lds9" / "global" with ``lds_exit`` saying which residency limit pushed the trees out of LDS, ``spec``, the workgroup shape,
        ``team_fallbacks``, ``last_ms`` and the kernel's name."""
        return self.engine.search_info()

    def results(self):
        return self.engine.results()

    def root_children(self):
        return self.engine.root_children()

    def close(self):
        self.engine.close()


class _Root:
    """What remains of the reference's root Node object on the host: the visit count a reused root carries."""

    def __init__(self, n: int, state):
        self.n = n
        self.state = state
        self.terminal = False
        self.parent_action = None


class MCTS:
    """Shared surface of mcts.py:23-307."""

    _mode = None

    def __init__(self, model, n_rollouts: int, c_uct: float, gamma: float, epsilon: float, device: str, V_target_policy: str,
                 root_state: np.ndarray, seed: int = 34):
        self.device = device
        self.root_node = None
        self.root_state = root_state
        self.model = model
        self.n_rollouts = n_rollouts
        self.c_uct = c_uct
        self.gamma = gamma
        self.epsilon = epsilon
        self.V_target_policy = V_target_policy
        self.seed = seed
        self._batched: Optional[BatchedMCTS] = None
        self._key = None
        self._res = None
        self._children = None
        self._envs: List[Any] = []

    # ---- engine management
    def _engine_kwargs(self) -> dict:
        raise NotImplementedError

    def _ensure_engine(self, env_id: int, n_trees: int) -> BatchedMCTS:
        device_id = device_ordinal(self.device)
        key = (env_id, n_trees, self.n_rollouts, self.c_uct, self.gamma, self.epsilon, self.V_target_policy, id(self.model),
               device_id, tuple(sorted(self._engine_kwargs().items())))
        if self._batched is None or key != self._key:
            if self._batched is not None:
                self._batched.close()
            self._batched = BatchedMCTS(self.model, env_id=env_id, mode=self._mode, n_trees=n_trees, n_rollouts=self.n_rollouts,
                                        c_uct=self.c_uct, gamma=self.gamma, epsilon=self.epsilon,
                                        V_target_policy=self.V_target_policy, seed=self.seed, device_id=device_id,
                                        **self._engine_kwargs())
            self._key = key
        return self._batched

    def _carry(self, n_trees: int) -> Optional[np.ndarray]:
        return None

    # ---- the reference's interface
    def search(self, Env) -> None:
        """Run n_rollouts traces from the state of ``Env`` (mcts.py:418-462 / 656-702).  ``Env`` is not mutated.
        A list/tuple of environments searches one tree per environment in the same launch."""
        envs: Sequence[Any] = list(Env) if isinstance(Env, (list, tuple)) else [Env]
        sigs = [env_signature(e) for e in envs]
        env_id = sigs[0][0]
        if any(s[0] != env_id for s in sigs):
            raise ValueError("all environments of one batched search must be of the same kind")
        roots = np.stack([s[1] for s in sigs])
        eng = self._ensure_engine(env_id, len(envs))
        eng.search(roots, self._carry(len(envs)))   # raises ValueError on a terminal root (mcts.py:382-383, 599-600)
        self._envs = list(envs)
        self._res = eng.results()
        self._children = eng.root_children()
        carried = 0 if self.root_node is None else self.root_node.n
        self.root_node = _Root(carried + self.n_rollouts, self.root_state)

    def _row(self, i: int):
        raise NotImplementedError

    def return_results(self, final_selection: str):
        """(root state, actions, counts, Q, V_target) with the reference's shapes and dtypes (mcts.py:269-307).
        After a batched search, a list with one such tuple per environment."""
        assert self._res is not None, "search() has not been called"
        rows = [self._row(i) for i in range(len(self._envs))]
        return rows[0] if len(rows) == 1 else rows

    # ---- value targets kept for API parity (mcts.py:92-131); the engine computes them on the device
    @staticmethod
    def get_on_policy_value_target(Q: np.ndarray, counts: np.ndarray) -> np.ndarray:
        return np.sum((counts / np.sum(counts)) * Q)

    @staticmethod
    def get_off_policy_value_target(Q: np.ndarray) -> Any:
        return Q.max()


class MCTSDiscrete(MCTS):
    """mcts.py:310-526.  Tree "reuse" keeps only the root's visit count, exactly as in the reference."""

    _mode = _capi.MODE_DISCRETE

    def __init__(self, model, num_actions: int, n_rollouts: int, c_uct: float, gamma: float, epsilon: float, V_target_policy: str,
                 device: str, root_state: np.ndarray, seed: int = 34):
        super().__init__(model=model, n_rollouts=n_rollouts, c_uct=c_uct, gamma=gamma, epsilon=epsilon, device=device,
                         V_target_policy=V_target_policy, root_state=root_state, seed=seed)
        self.num_actions = num_actions

    def _engine_kwargs(self) -> dict:
        return dict(num_actions=self.num_actions)

    def _carry(self, n_trees: int):
        if self.root_node is None:
            return None
        return np.full((n_trees,), int(self.root_node.n), dtype=np.int32)

    def _row(self, i: int):
        r = self._res
        k = int(r["n_children"][i])
        state = self.root_state if (len(self._envs) == 1 and self.root_state is not None) else env_observation(
            *env_signature(self._envs[i]))
        return (state, np.arange(k), r["counts"][i, :k].astype(np.int64), r["Q"][i, :k].copy(), float(r["v_target"][i]))

    def forward(self, action: int, state: np.ndarray) -> None:
        """Move the root to the child reached by ``action`` (mcts.py:495-526)."""
        assert self._children is not None
        child_n, child_state = self._children
        if child_n[0, action] < 0:
            self.root_node = None
            self.root_state = state
        elif np.linalg.norm(env_observation(env_signature(self._envs[0])[0], child_state[0, action])
                            - np.asarray(state, dtype=np.float32)) > 0.01:
            print("Warning: this domain seems stochastic. Not re-using the subtree for next search. "
                  + "To deal with stochastic environments, implement progressive widening.")
            self.root_node = None
            self.root_state = state
        else:
            self.root_node = _Root(int(child_n[0, action]), state)
            self.root_state = state


class MCTSContinuous(MCTS):
    """mcts.py:529-741: progressive widening; a new root every search (no tree reuse).  Nodes of environments whose episodes end
    (MountainCarContinuous) are terminal as in mcts.py:619-623, 682: value 0, no evaluation, the trace stops there; a terminal root
    state raises ValueError (mcts.py:599-600)."""

    _mode = _capi.MODE_CONTINUOUS

    def __init__(self, model, n_rollouts: int, c_uct: float, c_pw: float, kappa: float, gamma: float, epsilon: float,
                 V_target_policy: str, device: str, root_state: np.ndarray, seed: int = 34):
        super().__init__(model=model, n_rollouts=n_rollouts, c_uct=c_uct, gamma=gamma, epsilon=epsilon, device=device,
                         V_target_policy=V_target_policy, root_state=root_state, seed=seed)
        self.c_pw = c_pw
        self.kappa = kappa

    def _engine_kwargs(self) -> dict:
        bound = getattr(self.model, "action_bound", None)
        if not bound:
            raise NotImplementedError("the engine samples squashed-Normal actions: the policy needs a finite action_bound")
        return dict(c_pw=self.c_pw, kappa=self.kappa, action_bound=float(bound))

    def search(self, Env) -> None:
        self.root_node = None   # initialize_search always builds a fresh root (mcts.py:589-600)
        super().search(Env)

    def _row(self, i: int):
        r = self._res
        k = int(r["n_children"][i])
        if len(self._envs) == 1 and self.root_state is not None:
            state = self.root_state
        else:
            env_id, st = env_signature(self._envs[i])
            if env_id == _capi.ENV_MOUNTAINCAR_CONT:
                state = np.array(st)                       # (position, velocity): the observation is the state
            else:
                th, thdot = st
                state = np.array([np.cos(th), np.sin(th), thdot])
        actions = r["actions"][i, :k].copy()
        if k == 1:
            actions = actions.reshape(())   # np.squeeze of a single action (mcts.py:307)
        return (state, actions, r["counts"][i, :k].astype(np.int64), r["Q"][i, :k].reshape(k, 1).copy(), np.float64(r["v_target"][i]))
