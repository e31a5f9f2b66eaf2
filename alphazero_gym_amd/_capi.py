"""ctypes binding of the C ABI declared in include/azgym.h.

``Engine`` wraps one ``azg_engine*``.  The symbol prefix is a parameter only so that the test-suite can
bind the CPU oracle (prefix ``azo_``) through the same code; the product always binds ``azg_`` from
``csrc/libazgym_hip.so`` (see ``_native.py``).
"""
import ctypes as C
import math

import numpy as np

ABI_VERSION = 1

AZG_OK = 0
AZG_E_INVALID = -1
AZG_E_TERMINAL_ROOT = -2
AZG_E_DEVICE = -3
AZG_E_STATE = -4
AZG_E_UNSUPPORTED = -5

ENV_CARTPOLE, ENV_PENDULUM_V0, ENV_PENDULUM_V1, ENV_MOUNTAINCAR, ENV_MOUNTAINCAR_CONT, ENV_ACROBOT = 0, 1, 2, 3, 4, 5
MODE_DISCRETE, MODE_CONTINUOUS = 0, 1
VT = {"off_policy": 0, "on_policy": 1, "greedy": 2}
TIE = {"first": 0, "random": 1}   # helpers.argmax on exactly equal scores: lowest index (parity) or a Philox-keyed uniform pick
ACT = {"relu": 0, "elu": 1, "leakyrelu": 2, "relu6": 3, "silu": 4, "swish": 4, "hardswish": 5}
_ACT_MODULES = {"ReLU": 0, "ELU": 1, "LeakyReLU": 2, "ReLU6": 3, "SiLU": 4, "Hardswish": 5}
MAX_HIDDEN = 8

PENDULUM_R_SCALE = 16.2736044  # alphazero/search/mcts.py:20


class AzgConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("device_id", C.c_int32),
        ("env_id", C.c_int32),
        ("mode", C.c_int32),
        ("n_trees", C.c_int32),
        ("n_sims", C.c_int32),
        ("num_actions", C.c_int32),
        ("v_target", C.c_int32),
        ("tree_id_base", C.c_int32),
        ("tie_break", C.c_int32),
        ("c_uct", C.c_double),
        ("gamma", C.c_double),
        ("epsilon", C.c_double),
        ("c_pw", C.c_double),
        ("kappa", C.c_double),
        ("reward_scale", C.c_double),
        ("action_bound", C.c_double),
        ("seed", C.c_uint64),
    ]


class AzgMlpDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("in_dim", C.c_int32),
        ("n_hidden", C.c_int32),
        ("hidden", C.c_int32 * MAX_HIDDEN),
        ("n_dist", C.c_int32),
        ("activation", C.c_int32),
        ("log_std_min", C.c_float),
        ("log_std_max", C.c_float),
        ("num_components", C.c_int32),
        ("layernorm", C.c_int32),
    ]


class AzgSelfplayConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("max_episode_length", C.c_int32),
        ("deterministic", C.c_int32),
        ("capacity_steps", C.c_int32),
        ("final_selection", C.c_int32),
        ("ring_mode", C.c_int32),
        ("temperature", C.c_double),
        ("agent_epsilon", C.c_double),
    ]


FINAL_SELECTION = {"max_visit": 0, "max_visits": 0, "max_value": 1}   # the reference's configs spell it both ways


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"azgym error {code}: {msg}")
        self.code = code


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


SYMBOLS = [
    "abi_version", "engine_create", "engine_destroy", "last_error", "set_weights", "set_weights_device", "set_search_index", "search",
    "results", "results_resident", "root_children", "root_eval", "dump_tree", "max_children", "max_records", "env_state_dim", "obs_dim",
    "synthetic_roots", "last_search_ms", "upload_roots", "search_resident", "sync",
    "selfplay_begin", "selfplay_begin_ex", "selfplay_step", "selfplay_row_len", "selfplay_rows", "selfplay_stats",
    "selfplay_ring", "selfplay_rows_device", "mlp_eval",
]


def bind(lib, prefix):
    """Resolve every entry point of include/azgym.h on ``lib`` and set its prototype."""
    f = {}
    for s in SYMBOLS:
        f[s] = getattr(lib, prefix + s)
    vp = C.c_void_p
    f["abi_version"].restype = C.c_int
    f["engine_create"].argtypes = [C.POINTER(AzgConfig), C.POINTER(vp)]
    f["engine_destroy"].argtypes = [vp]
    f["engine_destroy"].restype = None
    f["last_error"].argtypes = [vp]
    f["last_error"].restype = C.c_char_p
    f["set_weights"].argtypes = [vp, C.POINTER(AzgMlpDesc), C.POINTER(C.c_float), C.c_size_t]
    f["set_weights_device"].argtypes = [vp, C.POINTER(AzgMlpDesc), C.c_void_p, C.c_size_t]
    f["results_resident"].argtypes = [vp] + [C.POINTER(C.c_void_p)] * 5
    f["set_search_index"].argtypes = [vp, C.c_uint32]
    f["search"].argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    f["results"].argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    f["root_children"].argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    f["root_eval"].argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    f["dump_tree"].argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                               C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                               C.POINTER(C.c_float), C.POINTER(C.c_uint8)]
    for s in ("max_children", "max_records", "env_state_dim", "obs_dim"):
        f[s].argtypes = [vp]
    f["synthetic_roots"].argtypes = [vp, C.POINTER(C.c_double)]
    f["last_search_ms"].argtypes = [vp, C.POINTER(C.c_float)]
    f["upload_roots"].argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    f["search_resident"].argtypes = [vp]
    f["sync"].argtypes = [vp]
    f["selfplay_begin"].argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    f["selfplay_begin_ex"].argtypes = [vp, C.POINTER(AzgSelfplayConfig)]
    f["selfplay_ring"].argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    f["selfplay_rows_device"].argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    f["selfplay_step"].argtypes = [vp]
    f["selfplay_row_len"].argtypes = [vp]
    f["selfplay_rows"].argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.c_int32]
    f["selfplay_stats"].argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    f["mlp_eval"].argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return f


def policy_tensors(policy):
    """(AzgMlpDesc, [tensors in blob order]) of a torch policy: per trunk layer weight, bias (, LayerNorm weight, bias), then the
    value head and the distribution head (= state_dict order).  Works for this package's policies and for the reference's
    DiscretePolicy / DiagonalNormalPolicy objects alike (same attribute names: trunk, value_head, dist_head, hidden_dimensions,
    state_dim; alphazero/network/policies.py:101-120, 238-259)."""
    hidden = list(policy.hidden_dimensions)
    linears, norms, acts = [], [], set()
    for mod in policy.trunk:
        name = type(mod).__name__
        if name == "Linear":
            linears.append(mod)
        elif name == "LayerNorm":
            norms.append(mod)
        elif name in _ACT_MODULES:
            acts.add(_ACT_MODULES[name])
        else:
            raise NotImplementedError(f"trunk module {name}: the engine implements Linear + activation (+ LayerNorm) trunks")
    if len(acts) != 1 or len(linears) != len(hidden) or len(norms) not in (0, len(hidden)):
        raise NotImplementedError("unsupported trunk structure")
    if any(abs(n.eps - 1e-5) > 1e-12 or not n.elementwise_affine for n in norms):
        raise NotImplementedError("LayerNorm must use eps=1e-5 and elementwise_affine=True (the torch defaults)")
    desc = AzgMlpDesc()
    desc.struct_size = C.sizeof(AzgMlpDesc)
    desc.in_dim = policy.state_dim
    desc.n_hidden = len(hidden)
    for i, h in enumerate(hidden):
        desc.hidden[i] = h
    desc.n_dist = policy.dist_head.out_features
    desc.activation = acts.pop()
    desc.log_std_min = float(getattr(policy, "log_param_min", -5.0))
    desc.log_std_max = float(getattr(policy, "log_param_max", 2.0))
    desc.num_components = int(getattr(policy, "num_components", 0) or 0)
    desc.layernorm = 1 if norms else 0
    tensors = []
    for i, mod in enumerate(linears):
        tensors += [mod.weight, mod.bias]
        if norms:
            tensors += [norms[i].weight, norms[i].bias]
    for mod in (policy.value_head, policy.dist_head):
        tensors += [mod.weight, mod.bias]
    return desc, tensors


def policy_blob(policy):
    """Flatten a torch policy into (AzgMlpDesc, float32 host blob) in state_dict order (see policy_tensors)."""
    desc, tensors = policy_tensors(policy)
    blob = np.ascontiguousarray(np.concatenate([t.detach().cpu().numpy().ravel() for t in tensors]), dtype=np.float32)
    return desc, blob


def make_desc(in_dim, hidden, n_dist, activation, log_std_min=-5.0, log_std_max=2.0, num_components=0, layernorm=False):
    desc = AzgMlpDesc()
    desc.struct_size = C.sizeof(AzgMlpDesc)
    desc.in_dim = in_dim
    desc.n_hidden = len(hidden)
    for i, h in enumerate(hidden):
        desc.hidden[i] = h
    desc.n_dist = n_dist
    desc.activation = ACT[activation] if isinstance(activation, str) else activation
    desc.log_std_min = log_std_min
    desc.log_std_max = log_std_max
    desc.num_components = num_components
    desc.layernorm = int(bool(layernorm))
    return desc


class Engine:
    """One batched MCTS engine (B trees).  Mirrors MCTS*.__init__ kwargs (alphazero/search/mcts.py:316-327, 537-549)."""

    def __init__(self, fns, *, env_id, mode, n_trees, n_sims, c_uct, gamma, epsilon=0.0, num_actions=0, c_pw=1.0,
                 kappa=0.5, v_target="off_policy", reward_scale=PENDULUM_R_SCALE, action_bound=2.0, seed=34,
                 tree_id_base=0, device_id=0, tie_break="first"):
        self._f = fns
        self._h = C.c_void_p()
        cfg = AzgConfig()
        cfg.struct_size = C.sizeof(AzgConfig)
        cfg.device_id = device_id
        cfg.env_id = env_id
        cfg.mode = mode
        cfg.n_trees = n_trees
        cfg.n_sims = n_sims
        cfg.num_actions = num_actions
        cfg.v_target = VT[v_target] if isinstance(v_target, str) else v_target
        cfg.tree_id_base = tree_id_base
        cfg.tie_break = TIE[tie_break] if isinstance(tie_break, str) else int(tie_break)
        cfg.c_uct = float(c_uct)
        cfg.gamma = float(gamma)
        cfg.epsilon = float(epsilon)
        cfg.c_pw = float(c_pw)
        cfg.kappa = float(kappa)
        cfg.reward_scale = float(reward_scale)
        cfg.action_bound = float(action_bound)
        cfg.seed = int(seed)
        self.cfg = cfg
        rc = fns["engine_create"](C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise EngineError(rc, (fns["last_error"](None) or b"").decode())
        self.n_trees = n_trees
        self.n_sims = n_sims
        self.mode = mode
        self.kmax = fns["max_children"](self._h)
        self.max_records = fns["max_records"](self._h)
        self.s_env = fns["env_state_dim"](self._h)
        self.s_obs = fns["obs_dim"](self._h)
        self.n_dist = num_actions if mode == MODE_DISCRETE else 2   # 3 * num_components after set_weights with a mixture head

    def _check(self, rc):
        if rc != 0:
            msg = (self._f["last_error"](self._h) or b"").decode()
            if rc == AZG_E_TERMINAL_ROOT:
                raise ValueError(msg or "Can't do tree search from a terminal node")
            raise EngineError(rc, msg)

    def close(self):
        if self._h:
            self._f["engine_destroy"](self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, desc, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._check(self._f["set_weights"](self._h, C.byref(desc), _ptr(blob, C.c_float), blob.size))
        if self.mode == MODE_CONTINUOUS:
            self.n_dist = desc.n_dist

    def set_weights_device(self, desc, device_ptr, n_floats):
        """azg_set_weights_device: the flat float32 blob already lives on the engine's GPU (complete: producer stream synchronised)."""
        self._check(self._f["set_weights_device"](self._h, C.byref(desc), C.c_void_p(int(device_ptr)), int(n_floats)))
        if self.mode == MODE_CONTINUOUS:
            self.n_dist = desc.n_dist

    def set_policy(self, policy):
        """Push a torch policy's weights.  Parameters on the engine's GPU are flattened there (one torch.cat into a device
        buffer) and re-laid-out by the engine's gather kernel: nothing crosses PCIe; CPU parameters take the host path."""
        par = next(policy.parameters(), None)
        if par is not None and par.is_cuda and par.device.index == self.cfg.device_id:
            import torch
            desc, tensors = policy_tensors(policy)
            with torch.no_grad():
                flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
            torch.cuda.current_stream(par.device).synchronize()
            self.set_weights_device(desc, flat.data_ptr(), flat.numel())
            return
        desc, blob = policy_blob(policy)
        self.set_weights(desc, blob)

    def results_resident(self):
        """azg_results_resident: launch return_results into the engine's device buffers; their addresses as a dict of ints."""
        ptrs = [C.c_void_p() for _ in range(5)]
        self._check(self._f["results_resident"](self._h, *[C.byref(p) for p in ptrs]))
        return dict(zip(("actions", "counts", "Q", "v_target", "n_children"), (p.value for p in ptrs)))

    def set_search_index(self, idx):
        self._check(self._f["set_search_index"](self._h, int(idx)))

    def _roots(self, roots, carry):
        roots = np.ascontiguousarray(roots, dtype=np.float64).reshape(self.n_trees, self.s_env)
        if carry is not None:
            carry = np.ascontiguousarray(carry, dtype=np.int32).reshape(self.n_trees)
        return roots, carry

    def search(self, roots, carry=None):
        roots, carry = self._roots(roots, carry)
        self._check(self._f["search"](self._h, _ptr(roots, C.c_double), _ptr(carry, C.c_int32)))

    def upload_roots(self, roots, carry=None):
        roots, carry = self._roots(roots, carry)
        self._check(self._f["upload_roots"](self._h, _ptr(roots, C.c_double), _ptr(carry, C.c_int32)))

    def search_resident(self):
        self._check(self._f["search_resident"](self._h))

    def sync(self):
        self._check(self._f["sync"](self._h))

    def last_search_ms(self):
        ms = C.c_float()
        self._check(self._f["last_search_ms"](self._h, C.byref(ms)))
        return ms.value

    def synthetic_roots(self):
        roots = np.empty((self.n_trees, self.s_env), dtype=np.float64)
        self._check(self._f["synthetic_roots"](self._h, _ptr(roots, C.c_double)))
        return roots

    def results(self):
        B, K = self.n_trees, self.kmax
        actions = np.empty((B, K), np.float32)
        counts = np.empty((B, K), np.int32)
        Q = np.empty((B, K), np.float64)
        vt = np.empty((B,), np.float64)
        nc = np.empty((B,), np.int32)
        self._check(self._f["results"](self._h, _ptr(actions, C.c_float), _ptr(counts, C.c_int32), _ptr(Q, C.c_double),
                                       _ptr(vt, C.c_double), _ptr(nc, C.c_int32)))
        return {"actions": actions, "counts": counts, "Q": Q, "v_target": vt, "n_children": nc}

    def root_children(self):
        B, K = self.n_trees, self.kmax
        child_n = np.empty((B, K), np.int32)
        child_state = np.empty((B, K, self.s_env), np.float64)
        self._check(self._f["root_children"](self._h, _ptr(child_n, C.c_int32), _ptr(child_state, C.c_double)))
        return child_n, child_state

    def root_eval(self):
        value = np.empty((self.n_trees,), np.float32)
        dist = np.empty((self.n_trees, self.n_dist), np.float32)
        self._check(self._f["root_eval"](self._h, _ptr(value, C.c_float), _ptr(dist, C.c_float)))
        return value, dist

    def mlp_eval(self, obs):
        """Batched network inference (policies.py predict_V / predict_pi / forward) of observations [n, obs_dim] with the
        search's own arithmetic: (value [n], dist [n, n_dist], raw head outputs [n, 1 + n_dist])."""
        obs = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1, self.s_obs)
        n = obs.shape[0]
        value = np.empty((n,), np.float32)
        dist = np.empty((n, self.n_dist), np.float32)
        raw = np.empty((n, 1 + self.n_dist), np.float32)
        self._check(self._f["mlp_eval"](self._h, _ptr(obs, C.c_float), n, _ptr(value, C.c_float), _ptr(dist, C.c_float),
                                        _ptr(raw, C.c_float)))
        return value, dist, raw

    def dump_tree(self):
        B, R = self.n_trees, self.max_records
        d = {
            "n_records": np.empty((B,), np.int32),
            "parent": np.empty((B, R), np.int32),
            "edge_n": np.empty((B, R), np.int32),
            "edge_W": np.empty((B, R), np.float64),
            "edge_Q": np.empty((B, R), np.float64),
            "edge_action": np.empty((B, R), np.float32),
            "node_n": np.empty((B, R), np.int32),
            "node_r": np.empty((B, R), np.float64),
            "node_V": np.empty((B, R), np.float32),
            "node_flags": np.empty((B, R), np.uint8),
        }
        self._check(self._f["dump_tree"](
            self._h, _ptr(d["n_records"], C.c_int32), _ptr(d["parent"], C.c_int32), _ptr(d["edge_n"], C.c_int32),
            _ptr(d["edge_W"], C.c_double), _ptr(d["edge_Q"], C.c_double), _ptr(d["edge_action"], C.c_float),
            _ptr(d["node_n"], C.c_int32), _ptr(d["node_r"], C.c_double), _ptr(d["node_V"], C.c_float),
            _ptr(d["node_flags"], C.c_uint8)))
        return d


def _selfplay_methods():
    def selfplay_begin(self, max_episode_length, deterministic=False, capacity_steps=64, final_selection="max_visit",
                       temperature=1.0, agent_epsilon=0.0, fifo=False):
        """Start device-resident self-play: games reset to their fixed-seed initial states (include/azgym.h).
        final_selection / temperature / agent_epsilon: the agents' final action rule (agents.py:294-301, 524-535);
        fifo: the ring overwrites its oldest step like ReplayBuffer.store (buffers.py:75-82) instead of refusing when full."""
        c = AzgSelfplayConfig()
        c.struct_size = C.sizeof(AzgSelfplayConfig)
        c.max_episode_length = int(max_episode_length)
        c.deterministic = int(bool(deterministic))
        c.capacity_steps = int(capacity_steps)
        c.final_selection = FINAL_SELECTION[final_selection] if isinstance(final_selection, str) else int(final_selection)
        c.ring_mode = 1 if fifo else 0
        c.temperature = float(temperature)
        c.agent_epsilon = float(agent_epsilon)
        self._check(self._f["selfplay_begin_ex"](self._h, C.byref(c)))
        self._sp_cap = int(capacity_steps)

    def selfplay_ring(self):
        """(size, insert_index, total) of the replay ring in steps: ReplayBuffer.size / .insert_index (buffers.py:56-82)."""
        size, ins, tot = C.c_int32(), C.c_int32(), C.c_int64()
        self._check(self._f["selfplay_ring"](self._h, C.byref(size), C.byref(ins), C.byref(tot)))
        return size.value, ins.value, tot.value

    def selfplay_rows_device(self):
        """(device pointer, capacity in rows, row length) of the replay ring, for consumers on the same GPU."""
        ptr, cap, rl = C.c_void_p(), C.c_size_t(), C.c_size_t()
        self._check(self._f["selfplay_rows_device"](self._h, C.byref(ptr), C.byref(cap), C.byref(rl)))
        return ptr.value, cap.value, rl.value

    def selfplay_step(self):
        """One search + final action + env step + bookkeeping for all games, entirely on the device (asynchronous)."""
        self._check(self._f["selfplay_step"](self._h))

    def selfplay_rows(self, clear=True):
        """Replay rows [steps*B, row_len] float32 of the steps since the last clear: obs | actions[K] | counts[K] | Q[K] | V."""
        rl = self._f["selfplay_row_len"](self._h)
        rows = np.empty((self._sp_cap * self.n_trees, rl), np.float32)
        n = self._f["selfplay_rows"](self._h, _ptr(rows, C.c_float), rows.shape[0], int(bool(clear)))
        if n < 0:
            self._check(n)
        return rows[:n]

    def selfplay_clear(self):
        """ReplayBuffer.clear (buffers.py:56-60) for the device ring, without downloading anything."""
        n = self._f["selfplay_rows"](self._h, None, 0, 1)
        if n < 0:
            self._check(n)

    def selfplay_stats(self):
        fsum = np.empty((self.n_trees,), np.float64)
        fcnt = np.empty((self.n_trees,), np.int32)
        state = np.empty((self.n_trees, self.s_env), np.float64)
        self._check(self._f["selfplay_stats"](self._h, _ptr(fsum, C.c_double), _ptr(fcnt, C.c_int32), _ptr(state, C.c_double)))
        return fsum, fcnt, state

    for fn in (selfplay_begin, selfplay_ring, selfplay_rows_device, selfplay_step, selfplay_rows, selfplay_clear, selfplay_stats):
        setattr(Engine, fn.__name__, fn)


_selfplay_methods()


def pw_table(c_pw, kappa, n):
    """NodeContinuous.check_pw's threshold (alphazero/search/states.py:271-273) for visit counts 0..n-1."""
    return [math.ceil(c_pw * ((i + 1) ** kappa)) for i in range(n)]
