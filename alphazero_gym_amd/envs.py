"""Closed-form classic-control environments (numpy float64 restatement).

The reference gets its environments from the third-party ``gym`` package (``rl/make_game.py:49-68``,
``gym==0.19.0`` in requirements.txt:10), which is neither vendored in the reference nor installed here.
These classes restate the published CartPole / MountainCar / MountainCarContinuous / Acrobot / Pendulum dynamics with the gym 0.19 ``Env`` call
surface the reference's MCTS uses: ``copy.deepcopy(Env)`` + ``Env.step(action)`` returning
``(obs, reward, done, info)`` (alphazero/search/mcts.py:443-449, 680-687), ``Env.reset()``, ``Env.seed()``.

They are the definition of "the environment" for every parity claim in this repo (env
parity is *unpinned* against gym itself).  One deliberate choice: a float32 action is widened to
float64 before it enters the arithmetic, so the step is pure float64 under every NumPy version.
The device kernels and the C oracle implement the same operation order.
"""
import math

import numpy as np

ENV_CARTPOLE = 0
ENV_PENDULUM_V0 = 1
ENV_PENDULUM_V1 = 2
ENV_MOUNTAINCAR = 3
ENV_MOUNTAINCAR_CONT = 4
ENV_ACROBOT = 5


class _Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = dtype


class _Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()
        self.dtype = np.int64


class _EnvBase:
    azg_env_id = -1

    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def close(self):
        pass

    def azg_state(self):
        """float64 internal state vector handed to the engine as the search root."""
        return np.asarray(self.state, dtype=np.float64)


class CartPoleEnv(_EnvBase):
    """gym ``CartPole-v0/v1`` dynamics (explicit Euler); TimeLimit is stripped by make_game (rl/make_game.py:61-62)."""

    azg_env_id = ENV_CARTPOLE
    gravity = 9.8
    masscart = 1.0
    masspole = 0.1
    total_mass = masspole + masscart
    length = 0.5
    polemass_length = masspole * length
    force_mag = 10.0
    tau = 0.02
    theta_threshold_radians = 12 * 2 * math.pi / 360
    x_threshold = 2.4

    def __init__(self, state=None):
        high = np.array([self.x_threshold * 2, np.finfo(np.float32).max, self.theta_threshold_radians * 2, np.finfo(np.float32).max])
        self.observation_space = _Box(-high, high, (4,))
        self.action_space = _Discrete(2)
        self.seed(None)
        self.state = None if state is None else tuple(float(v) for v in state)

    def reset(self):
        self.state = tuple(float(v) for v in self.np_random.uniform(low=-0.05, high=0.05, size=(4,)))
        return np.array(self.state, dtype=np.float32)

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if int(action) == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + (self.polemass_length * (theta_dot * theta_dot)) * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - (self.masspole * (costheta * costheta)) / self.total_mass)
        )
        xacc = temp - ((self.polemass_length * thetaacc) * costheta) / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = (x, x_dot, theta, theta_dot)
        done = bool(
            x < -self.x_threshold or x > self.x_threshold or theta < -self.theta_threshold_radians or theta > self.theta_threshold_radians
        )
        return np.array(self.state, dtype=np.float32), 1.0, done, {}


class MountainCarEnv(_EnvBase):
    """gym ``MountainCar-v0`` (three discrete actions: push left, no push, push right); TimeLimit stripped like CartPole's.
    The arithmetic is gym 0.19's ``MountainCarEnv.step`` on python floats: ``velocity += (action - 1) * force +
    cos(3 * position) * (-gravity)``, clip, ``position += velocity``, clip, inelastic left wall, reward -1.0."""

    azg_env_id = ENV_MOUNTAINCAR
    min_position = -1.2
    max_position = 0.6
    max_speed = 0.07
    goal_position = 0.5
    goal_velocity = 0.0
    force = 0.001
    gravity = 0.0025

    def __init__(self, state=None):
        low = np.array([self.min_position, -self.max_speed], dtype=np.float32)
        high = np.array([self.max_position, self.max_speed], dtype=np.float32)
        self.observation_space = _Box(low, high, (2,))
        self.action_space = _Discrete(3)
        self.seed(None)
        self.state = None if state is None else tuple(float(v) for v in state)

    def reset(self):
        self.state = (float(self.np_random.uniform(low=-0.6, high=-0.4)), 0.0)
        return np.array(self.state, dtype=np.float32)

    def step(self, action):
        position, velocity = self.state
        velocity = velocity + ((int(action) - 1) * self.force + math.cos(3 * position) * (-self.gravity))
        velocity = min(max(velocity, -self.max_speed), self.max_speed)
        position = position + velocity
        position = min(max(position, self.min_position), self.max_position)
        if position == self.min_position and velocity < 0:
            velocity = 0.0
        done = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        self.state = (position, velocity)
        return np.array(self.state, dtype=np.float32), -1.0, done, {}


class AcrobotEnv(_EnvBase):
    """gym ``Acrobot-v1``: two links, three discrete torques (-1, 0, +1) on the joint between them, SIX observations
    ``(cos t1, sin t1, cos t2, sin t2, dt1, dt2)``, reward -1 per step and 0 on the step that ends the episode (the tip above the line:
    ``-cos t1 - cos(t1 + t2) > 1``).  gym 0.19's "book" dynamics integrated by one classical Runge-Kutta step of 0.2 s (its ``rk4``), then
    angles wrapped into [-pi, pi] and velocities bounded to +-4 pi / +-9 pi; no torque noise.  All masses, lengths and inertias are 1, the
    centres of mass sit at 0.5 and g = 9.8: the constants are multiplied out below in gym's own order of operations wherever a factor
    is not exactly 1, on python floats (gym works on numpy float64 scalars: the same arithmetic) -- this method is the definition the C
    oracle and the device kernels mirror (include/azg_math.h: azg_acrobot_dsdt / azg_acrobot_step)."""

    azg_env_id = ENV_ACROBOT
    dt = 0.2
    MAX_VEL_1 = 4 * math.pi
    MAX_VEL_2 = 9 * math.pi
    AVAIL_TORQUE = (-1.0, 0.0, +1.0)

    def __init__(self, state=None):
        high = np.array([1.0, 1.0, 1.0, 1.0, self.MAX_VEL_1, self.MAX_VEL_2], dtype=np.float32)
        self.observation_space = _Box(-high, high, (6,))
        self.action_space = _Discrete(3)
        self.seed(None)
        self.state = None if state is None else np.asarray(state, dtype=np.float64)

    def reset(self):
        self.state = self.np_random.uniform(low=-0.1, high=0.1, size=(4,))
        return self._get_ob()

    def _get_ob(self):
        s = self.state
        return np.array([math.cos(s[0]), math.sin(s[0]), math.cos(s[1]), math.sin(s[1]), s[2], s[3]])

    @staticmethod
    def _dsdt(s, a):
        g = 9.8
        theta1, theta2, dtheta1, dtheta2 = s
        cs2, sn2 = math.cos(theta2), math.sin(theta2)
        d1 = ((0.25 + (1.25 + cs2)) + 1.0) + 1.0
        d2 = (0.25 + 0.5 * cs2) + 1.0
        phi2 = (0.5 * g) * math.cos((theta1 + theta2) - math.pi / 2.0)
        phi1 = ((-(0.5 * (dtheta2 * dtheta2)) * sn2 - ((1.0 * dtheta2) * dtheta1) * sn2) + (1.5 * g) * math.cos(theta1 - math.pi / 2.0)) + phi2
        ddtheta2 = (((a + (d2 / d1) * phi1) - (0.5 * (dtheta1 * dtheta1)) * sn2) - phi2) / ((0.25 + 1.0) - (d2 * d2) / d1)
        ddtheta1 = -(d2 * ddtheta2 + phi1) / d1
        return (dtheta1, dtheta2, ddtheta1, ddtheta2)

    def _terminal(self):
        s = self.state
        return bool(-math.cos(s[0]) - math.cos(s[1] + s[0]) > 1.0)

    def step(self, action):
        s = tuple(float(v) for v in self.state)
        a = self.AVAIL_TORQUE[int(action)]
        dt, dt2 = self.dt, self.dt / 2.0
        k1 = self._dsdt(s, a)
        k2 = self._dsdt(tuple(s[i] + dt2 * k1[i] for i in range(4)), a)
        k3 = self._dsdt(tuple(s[i] + dt2 * k2[i] for i in range(4)), a)
        k4 = self._dsdt(tuple(s[i] + dt * k3[i] for i in range(4)), a)
        ns = [s[i] + (dt / 6.0) * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]) for i in range(4)]
        for i in range(2):   # wrap(x, -pi, pi)
            x = ns[i]
            while x > math.pi:
                x = x - 2.0 * math.pi
            while x < -math.pi:
                x = x + 2.0 * math.pi
            ns[i] = x
        ns[2] = min(max(ns[2], -self.MAX_VEL_1), self.MAX_VEL_1)
        ns[3] = min(max(ns[3], -self.MAX_VEL_2), self.MAX_VEL_2)
        self.state = np.array(ns)
        terminal = self._terminal()
        return self._get_ob(), (-1.0 if not terminal else 0.0), terminal, {}


class MountainCarContinuousEnv(_EnvBase):
    """gym ``MountainCarContinuous-v0`` (``Continuous_MountainCarEnv``): one continuous action in [-1, 1], episodes END at the flag
    (position >= 0.45 with velocity >= 0) -- the env through which the continuous search meets terminal nodes (mcts.py:619-623,
    682).  gym 0.19's step on python floats: ``force = min(max(action[0], -1), 1)``; ``velocity += force * power - 0.0025 *
    cos(3 * position)``, clipped to +-0.07; ``position += velocity``, clipped to [-1.2, 0.6]; inelastic wall on the left;
    ``reward = 100 * done - 0.1 * action[0] ** 2`` (the action as it came, not the clipped force).

    Two deliberate choices, as for Pendulum: the float32 action is widened to float64 before it enters the arithmetic, and the
    reward is returned as a float64 array of shape (1,) -- the same container PendulumEnv uses.  gym hands back a python float
    here; under NumPy >= 2 (NEP 50) a python float is "weak", ``node.r + gamma * R`` with a float32 ``R`` (mcts.py:262) would then
    be a float32 sum and every W / Q of the tree would silently live in float32.  The float64 array keeps the reference's
    statistics in float64 and Q in shape (K, 1), identical to what the Pendulum goldens pin."""

    azg_env_id = ENV_MOUNTAINCAR_CONT
    min_action = -1.0
    max_action = 1.0
    min_position = -1.2
    max_position = 0.6
    max_speed = 0.07
    goal_position = 0.45
    goal_velocity = 0.0
    power = 0.0015

    def __init__(self, state=None):
        low = np.array([self.min_position, -self.max_speed], dtype=np.float32)
        high = np.array([self.max_position, self.max_speed], dtype=np.float32)
        self.observation_space = _Box(low, high, (2,))
        self.action_space = _Box(self.min_action, self.max_action, (1,))
        self.seed(None)
        self.state = None if state is None else np.asarray(state, dtype=np.float64)

    def reset(self):
        self.state = np.array([self.np_random.uniform(low=-0.6, high=-0.4), 0.0])
        return np.array(self.state)

    def step(self, action):
        position, velocity = (float(v) for v in self.state)
        a = float(np.asarray(action, dtype=np.float32).reshape(-1)[0])
        force = min(max(a, self.min_action), self.max_action)
        velocity = velocity + (force * self.power - 0.0025 * math.cos(3 * position))
        if velocity > self.max_speed:
            velocity = self.max_speed
        if velocity < -self.max_speed:
            velocity = -self.max_speed
        position = position + velocity
        if position > self.max_position:
            position = self.max_position
        if position < self.min_position:
            position = self.min_position
        if position == self.min_position and velocity < 0:
            velocity = 0.0
        done = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        reward = (100.0 if done else 0.0) - (a * a) * 0.1
        self.state = np.array([position, velocity])
        return np.array(self.state), np.array([reward]), done, {}


class PendulumEnv(_EnvBase):
    """gym ``Pendulum-v0`` (``version=0``: speed clipped after integrating theta) / ``Pendulum-v1`` (clipped before)."""

    max_speed = 8.0
    max_torque = 2.0
    dt = 0.05
    g = 10.0
    m = 1.0
    l = 1.0

    def __init__(self, state=None, version=1):
        assert version in (0, 1)
        self.version = version
        self.azg_env_id = ENV_PENDULUM_V1 if version == 1 else ENV_PENDULUM_V0
        high = np.array([1.0, 1.0, self.max_speed], dtype=np.float32)
        self.observation_space = _Box(-high, high, (3,))
        self.action_space = _Box(-self.max_torque, self.max_torque, (1,))
        self.seed(None)
        self.state = None if state is None else np.asarray(state, dtype=np.float64)

    def reset(self):
        high = np.array([np.pi, 1.0])
        self.state = self.np_random.uniform(low=-high, high=high)
        return self._get_obs()

    def _get_obs(self):
        th, thdot = self.state
        return np.array([np.cos(th), np.sin(th), thdot])

    def step(self, u):
        th, thdot = (float(v) for v in self.state)
        uc = np.clip(np.asarray(u, dtype=np.float32).reshape(-1)[0], np.float32(-self.max_torque), np.float32(self.max_torque))
        u = float(uc)
        an = ((th + math.pi) % (2.0 * math.pi)) - math.pi
        costs = (an * an + 0.1 * (thdot * thdot)) + 0.001 * (u * u)
        if self.version == 1:
            newthdot = thdot + (15.0 * math.sin(th) + 3.0 * u) * self.dt
            newthdot = min(max(newthdot, -self.max_speed), self.max_speed)
            newth = th + newthdot * self.dt
        else:
            newthdot = thdot + (-15.0 * math.sin(th + math.pi) + 3.0 * u) * self.dt
            newth = th + newthdot * self.dt
            newthdot = min(max(newthdot, -self.max_speed), self.max_speed)
        self.state = np.array([newth, newthdot])
        # shape-(1,) reward like gym's when the action is a length-1 array (keeps the reference's Q shape (K,1))
        return self._get_obs(), np.array([-costs]), False, {}


def make_game(game: str):
    """Counterpart of rl/make_game.py:49-68 for the closed-form envs (no wrappers, no TimeLimit)."""
    name = game.split("-")[0].lower()
    if name == "cartpole":
        return CartPoleEnv()
    if name == "pendulum":
        return PendulumEnv(version=0 if game.endswith("v0") else 1)
    if name == "mountaincar":
        return MountainCarEnv()
    if name == "mountaincarcontinuous":
        return MountainCarContinuousEnv()
    if name == "acrobot":
        return AcrobotEnv()
    raise ValueError(f"unsupported game {game!r}: this engine ships closed-form CartPole, MountainCar(Continuous), Acrobot and Pendulum only")


class VecPendulum:
    """B Pendulum environments stepped together (numpy float64, same operation order as PendulumEnv.step)."""

    def __init__(self, n: int, version: int = 1, seed: int = 0):
        self.n, self.version = n, version
        self.azg_env_id = ENV_PENDULUM_V1 if version == 1 else ENV_PENDULUM_V0
        self.rng = np.random.RandomState(seed)
        self.state = np.zeros((n, 2))
        self.reset(np.ones(n, bool))

    def reset(self, mask):
        k = int(mask.sum())
        if k:
            self.state[mask] = self.rng.uniform(low=[-np.pi, -1.0], high=[np.pi, 1.0], size=(k, 2))

    def obs(self):
        th, thd = self.state[:, 0], self.state[:, 1]
        return np.stack([np.cos(th), np.sin(th), thd], 1)

    def step(self, u):
        th, thd = self.state[:, 0], self.state[:, 1]
        u = np.clip(np.asarray(u, np.float32).reshape(-1), np.float32(-2.0), np.float32(2.0)).astype(np.float64)
        an = ((th + np.pi) % (2.0 * np.pi)) - np.pi
        costs = (an * an + 0.1 * (thd * thd)) + 0.001 * (u * u)
        if self.version == 1:
            nthd = np.clip(thd + (15.0 * np.sin(th) + 3.0 * u) * 0.05, -8.0, 8.0)
            nth = th + nthd * 0.05
        else:
            nthd = thd + (-15.0 * np.sin(th + np.pi) + 3.0 * u) * 0.05
            nth = th + nthd * 0.05
            nthd = np.clip(nthd, -8.0, 8.0)
        self.state = np.stack([nth, nthd], 1)
        return -costs, np.zeros(self.n, bool)


class VecMountainCarContinuous:
    """B MountainCarContinuous environments stepped together (numpy float64, same operation order as MountainCarContinuousEnv.step)."""

    azg_env_id = ENV_MOUNTAINCAR_CONT

    def __init__(self, n: int, seed: int = 0):
        self.n = n
        self.rng = np.random.RandomState(seed)
        self.state = np.zeros((n, 2))
        self.reset(np.ones(n, bool))

    def reset(self, mask):
        k = int(mask.sum())
        if k:
            self.state[mask] = np.stack([self.rng.uniform(low=-0.6, high=-0.4, size=k), np.zeros(k)], 1)

    def obs(self):
        return self.state.astype(np.float32)

    def step(self, action):
        x, v = self.state[:, 0], self.state[:, 1]
        a = np.asarray(action, np.float32).reshape(-1).astype(np.float64)
        force = np.minimum(np.maximum(a, -1.0), 1.0)
        v = v + (force * 0.0015 - 0.0025 * np.cos(3 * x))
        v = np.where(v > 0.07, 0.07, v)
        v = np.where(v < -0.07, -0.07, v)
        x = x + v
        x = np.where(x > 0.6, 0.6, x)
        x = np.where(x < -1.2, -1.2, x)
        v = np.where((x == -1.2) & (v < 0), 0.0, v)
        done = (x >= 0.45) & (v >= 0.0)
        self.state = np.stack([x, v], 1)
        return np.where(done, 100.0, 0.0) - (a * a) * 0.1, done


class VecCartPole:
    """B CartPole environments stepped together (numpy float64, same operation order as CartPoleEnv.step)."""

    azg_env_id = ENV_CARTPOLE

    def __init__(self, n: int, seed: int = 0):
        self.n = n
        self.rng = np.random.RandomState(seed)
        self.state = np.zeros((n, 4))
        self.reset(np.ones(n, bool))

    def reset(self, mask):
        k = int(mask.sum())
        if k:
            self.state[mask] = self.rng.uniform(low=-0.05, high=0.05, size=(k, 4))

    def obs(self):
        return self.state.astype(np.float32)

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state.T
        force = np.where(np.asarray(action).reshape(-1) == 1, 10.0, -10.0)
        costheta, sintheta = np.cos(theta), np.sin(theta)
        total_mass, pml = CartPoleEnv.total_mass, CartPoleEnv.polemass_length
        temp = (force + (pml * (theta_dot * theta_dot)) * sintheta) / total_mass
        thetaacc = (9.8 * sintheta - costheta * temp) / (0.5 * (4.0 / 3.0 - (0.1 * (costheta * costheta)) / total_mass))
        xacc = temp - ((pml * thetaacc) * costheta) / total_mass
        x = x + 0.02 * x_dot
        x_dot = x_dot + 0.02 * xacc
        theta = theta + 0.02 * theta_dot
        theta_dot = theta_dot + 0.02 * thetaacc
        self.state = np.stack([x, x_dot, theta, theta_dot], 1)
        thr = CartPoleEnv.theta_threshold_radians
        done = (x < -2.4) | (x > 2.4) | (theta < -thr) | (theta > thr)
        return np.ones(self.n), done
