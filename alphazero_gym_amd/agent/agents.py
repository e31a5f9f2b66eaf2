"""Agents with the reference's surface (alphazero/agent/agents.py): ``act`` = one engine search + the final action
rule, ``update``/``train`` = the PyTorch optimiser step over replay-buffer minibatches, ``reset_mcts``,
``mcts_forward`` and the read-only properties the run scripts log."""
import random
from abc import ABC, abstractmethod
from collections import defaultdict
from typing import Any, Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ..config import instantiate
from ..helpers import stable_normalizer
from ..search.mcts import MCTSContinuous, MCTSDiscrete
from .buffers import ReplayBuffer
from .losses import A0CLoss

Obs = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]


class Agent(ABC):
    """agents.py:19-184.  The four *_cfg arguments are `_target_` mappings (hydra DictConfigs work: they are Mappings)
    or already-built objects."""

    def __init__(self, policy_cfg, mcts_cfg, loss_cfg, optimizer_cfg, final_selection: str, train_epochs: int,
                 grad_clip: float, device: str) -> None:
        self.device = torch.device(device)
        self.nn = instantiate(policy_cfg).to(self.device)
        self.mcts = instantiate(mcts_cfg, model=self.nn)
        loss = instantiate(loss_cfg)
        self.loss = loss.to(self.device) if hasattr(loss, "to") else loss
        self.optimizer = instantiate(optimizer_cfg, params=self.nn.parameters())
        self.final_selection = final_selection
        self.train_epochs = train_epochs
        self.clip = grad_clip

    @abstractmethod
    def act(self, Env): ...

    @abstractmethod
    def _loss(self, states, actions, counts, values) -> Dict[str, Any]:
        """Loss dictionary of one minibatch already on ``self.device`` (float32 tensors; values is [n, 1])."""

    def _tensor(self, x):
        """A batch field on the training device as float32: numpy arrays from ReplayBuffer, or tensors that already live on
        the GPU (DeviceReplay minibatches: no host round trip)."""
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        return t.to(self.device, dtype=torch.float32, non_blocking=True)

    def update(self, obs: Obs) -> Dict[str, float]:
        """One optimiser step on a minibatch (states, actions, counts, Qs, V_target) (agents.py:319-392, 539-603)."""
        states, actions, counts, _, v_target = obs
        self.optimizer.zero_grad(set_to_none=True)
        loss_dict = self._loss(self._tensor(states), self._tensor(actions), self._tensor(counts), self._tensor(v_target).reshape(-1, 1))
        loss_dict["loss"].backward()
        if self.clip:
            torch.nn.utils.clip_grad_norm_(self.nn.parameters(), self.clip)
        self.optimizer.step()
        return {key: float(value.detach()) if hasattr(value, "detach") else float(value) for key, value in loss_dict.items()}

    # read-only views the run scripts log (agents.py:106-144): network shape from the policy, search settings from the MCTS object
    action_dim = property(lambda self: self.nn.action_dim)
    state_dim = property(lambda self: self.nn.state_dim)
    n_hidden_layers = property(lambda self: self.nn.n_hidden_layers)
    n_hidden_units = property(lambda self: self.nn.n_hidden_units)
    n_rollouts = property(lambda self: self.mcts.n_rollouts)
    c_uct = property(lambda self: self.mcts.c_uct)
    gamma = property(lambda self: self.mcts.gamma)
    learning_rate = property(lambda self: self.optimizer.param_groups[0]["lr"])

    def reset_mcts(self, root_state: np.ndarray) -> None:
        """agents.py:146-155"""
        self.mcts.root_node = None
        self.mcts.root_state = root_state

    def train(self, buffer: ReplayBuffer) -> Dict[str, Any]:
        """One pass of minibatch updates per epoch over the reshuffled buffer (agents.py:157-184); like the reference,
        the per-key sums are returned (its division by the batch count has no effect)."""
        buffer.reshuffle()
        running_loss: Dict[str, Any] = defaultdict(float)
        for _ in range(self.train_epochs):
            for obs in buffer:
                loss = self.update(obs)
                for key, val in loss.items():
                    running_loss[key] += val
        return running_loss


class DiscreteAgent(Agent):
    """agents.py:187-392"""

    def __init__(self, policy_cfg, mcts_cfg, loss_cfg, optimizer_cfg, final_selection: str, train_epochs: int, grad_clip: float,
                 temperature: float, device: str) -> None:
        super().__init__(policy_cfg=policy_cfg, loss_cfg=loss_cfg, mcts_cfg=mcts_cfg, optimizer_cfg=optimizer_cfg,
                         final_selection=final_selection, train_epochs=train_epochs, grad_clip=grad_clip, device=device)
        assert isinstance(self.mcts, MCTSDiscrete)
        self.temperature = temperature

    def act(self, Env, deterministic: bool = False):
        """Search, then sample the action from the (temperature-scaled) visit-count or Q distribution (agents.py:257-303).
        Returns (action, state, actions, counts, Qs, V)."""
        self.mcts.search(Env=Env)
        state, actions, counts, Qs, V = self.mcts.return_results(self.final_selection)
        pi = stable_normalizer(Qs if self.final_selection == "max_value" else counts, self.temperature)
        action = pi.argmax() if deterministic else np.random.choice(len(pi), p=pi)
        return action, state, actions, counts, Qs, V

    def mcts_forward(self, action: int, node: np.ndarray) -> None:
        self.mcts.forward(action, node)

    def _loss(self, states, actions, counts, values):
        if isinstance(self.loss, A0CLoss):
            # the A0C losses on a discrete policy; counts + 1 keeps log(counts) finite (agents.py:364)
            log_probs, entropy, V_hat = self.nn.get_train_data(states, actions)
            return self.loss(log_probs=log_probs, counts=counts + 1, entropy=entropy, V=values, V_hat=V_hat)
        # AlphaZero loss: cross-entropy against softmax(counts) (agents.py:378-380)
        dist, V_hat = self.nn(states)
        return self.loss(dist.logits, F.softmax(counts, dim=-1), V_hat, values)


class ContinuousAgent(Agent):
    """agents.py:395-603"""

    def __init__(self, policy_cfg, mcts_cfg, loss_cfg, optimizer_cfg, final_selection: str, epsilon: float, train_epochs: int,
                 grad_clip: float, device: str) -> None:
        super().__init__(policy_cfg=policy_cfg, loss_cfg=loss_cfg, mcts_cfg=mcts_cfg, optimizer_cfg=optimizer_cfg,
                         final_selection=final_selection, train_epochs=train_epochs, grad_clip=grad_clip, device=device)
        assert isinstance(self.mcts, MCTSContinuous)
        self.epsilon = epsilon

    @property
    def action_limit(self) -> float:
        return self.nn.action_bound

    def epsilon_greedy(self, actions: np.ndarray, values: np.ndarray) -> np.ndarray:
        """agents.py:471-490"""
        if random.random() < self.epsilon:
            return np.random.choice(actions)[np.newaxis]
        return actions[values.argmax()][np.newaxis]

    def act(self, Env):
        """Search, then the most visited (or highest-Q) root action, first index on ties (agents.py:492-537)."""
        self.mcts.search(Env=Env)
        state, actions, counts, Qs, V = self.mcts.return_results(self.final_selection)
        values = Qs if self.final_selection == "max_value" else counts
        actions1 = np.atleast_1d(actions)
        if self.epsilon == 0:
            action = actions1[values.argmax()][np.newaxis]
        else:
            action = self.epsilon_greedy(actions=actions1, values=values)
        return action, state, actions, counts, Qs, V

    def _loss(self, states, actions, counts, values):
        log_probs, entropy, V_hat = self.nn.get_train_data(states, actions)
        return self.loss(log_probs=log_probs, counts=counts, entropy=entropy, V=values, V_hat=V_hat)
