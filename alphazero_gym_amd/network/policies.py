"""Policy/value networks for the optimiser step (PyTorch) with the reference's constructor arguments, attribute names
and inference methods (alphazero/network/policies.py).  The search itself never calls these modules: the engine
reads their weights (alphazero_gym_amd._capi.policy_blob) and evaluates the MLP on the GPU's matrix cores.

Supported: DiscretePolicy (policies.py:163-352), DiagonalNormalPolicy (policies.py:355-499) and DiagonalGMMPolicy
(policies.py:502-669, the reference's default continuous head).  The Beta head (policies.py:672-803) is not built (the
reference marks it as not working, README.md:21-22)."""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .distributions import SquashedNormal

_ACTIVATIONS = {"relu": nn.ReLU, "elu": nn.ELU, "leakyrelu": nn.LeakyReLU, "relu6": nn.ReLU6, "silu": nn.SiLU, "swish": nn.SiLU,
                "hardswish": nn.Hardswish}   # alphazero/network/utils.py:5-14


def _trunk(in_dim: int, hidden: List[int], nonlinearity: str, layernorm: bool) -> nn.Sequential:
    key = nonlinearity.lower().replace(" ", "")
    if key not in _ACTIVATIONS:
        raise NotImplementedError(f"nonlinearity {nonlinearity!r}: one of {sorted(_ACTIVATIONS)}")
    layers, k = [], in_dim
    for h in hidden:
        layers += [nn.Linear(k, h), _ACTIVATIONS[key]()]
        if layernorm:
            layers.append(nn.LayerNorm(normalized_shape=h))   # after the activation, like the reference (policies.py:105-106)
        k = h
    return nn.Sequential(*layers)


class Policy(nn.Module):
    """Common part of the policy networks (the reference's base class of the same name, policies.py:22-120): trunk + value head."""

    def _setup(self, representation_dim, action_dim, hidden_dimensions, nonlinearity, layernorm):
        assert hidden_dimensions, "Hidden dimensions can't be empty."
        self.state_dim = representation_dim
        self.action_dim = action_dim
        self.hidden_dimensions = list(hidden_dimensions)
        self.hidden_layers = len(hidden_dimensions)
        self.nonlinearity = nonlinearity.lower().replace(" ", "")
        self.layernorm = layernorm
        self.trunk = _trunk(representation_dim, self.hidden_dimensions, nonlinearity, layernorm)
        self.value_head = nn.Linear(self.hidden_dimensions[-1], 1)

    @property
    def n_hidden_layers(self) -> int:
        return self.hidden_layers

    @property
    def n_hidden_units(self) -> int:
        return sum(self.hidden_dimensions)


class DiscretePolicy(Policy):
    distribution_type = "Categorical"

    def __init__(self, representation_dim: int, action_dim: int, num_actions: int, hidden_dimensions: List[int],
                 nonlinearity: str, layernorm: bool = False):
        super().__init__()
        self._setup(representation_dim, action_dim, hidden_dimensions, nonlinearity, layernorm)
        self.num_actions = num_actions
        self.dist_head = nn.Linear(self.hidden_dimensions[-1], num_actions)

    def _get_dist_params(self, x):
        h = self.trunk(x)
        return self.dist_head(h), self.value_head(h)

    def forward(self, x):
        logits, V_hat = self._get_dist_params(x)
        return torch.distributions.Categorical(logits=logits), V_hat

    def get_train_data(self, states, actions):
        """log-probs [B, num_actions], entropy [B, num_actions], V_hat [B, 1] (policies.py:319-338)."""
        logits, V_hat = self._get_dist_params(states)
        num_actions = actions.shape[1]
        pi_hat = torch.distributions.Categorical(logits=logits.unsqueeze(dim=1).repeat((1, num_actions, 1)))
        return pi_hat.log_prob(actions), pi_hat.entropy(), V_hat

    @torch.no_grad()
    def predict_V(self, x) -> np.ndarray:
        _, V_hat = self._get_dist_params(x)
        return V_hat.detach().cpu().numpy()

    @torch.no_grad()
    def predict_pi(self, x) -> np.ndarray:
        logits, _ = self._get_dist_params(x)
        return F.softmax(logits, dim=-1).detach().cpu().numpy()


class DiagonalNormalPolicy(Policy):
    distribution_type = "Normal"
    policy_type = "DiagonalNormal"

    def __init__(self, representation_dim: int, action_dim: int, action_bound: Optional[float], hidden_dimensions: List[int],
                 nonlinearity: str, layernorm: bool = False, log_param_min: float = -5, log_param_max: float = 2):
        super().__init__()
        self._setup(representation_dim, action_dim, hidden_dimensions, nonlinearity, layernorm)
        self.action_bound = action_bound
        self.log_param_min = log_param_min
        self.log_param_max = log_param_max
        self.dist_head = nn.Linear(self.hidden_dimensions[-1], 2 * action_dim)

    @property
    def bounds(self) -> np.ndarray:
        if self.action_bound is None:
            return np.array([-np.inf, np.inf], dtype=np.float32)
        return np.array([-self.action_bound, self.action_bound], dtype=np.float32)

    def forward(self, x):
        h = self.trunk(x)
        V_hat = self.value_head(h)
        mu, log_std = self.dist_head(h).chunk(2, dim=-1)
        log_std = torch.clamp(log_std, min=self.log_param_min, max=self.log_param_max)
        return mu, log_std.exp(), V_hat

    def _dist(self, mu, sigma):
        if self.action_bound:
            return SquashedNormal(mu, sigma, self.action_bound)
        return torch.distributions.Normal(mu, sigma)

    def get_train_data(self, states, actions):
        """log-probs [B, K], entropy estimate [B], V_hat [B, 1] (policies.py:466-486)."""
        mu, sigma, V_hat = self(states)
        log_probs = self._dist(mu, sigma).log_prob(actions)
        return log_probs, -log_probs.mean(dim=-1), V_hat

    @torch.no_grad()
    def predict_V(self, x) -> np.ndarray:
        return self.value_head(self.trunk(x)).detach().cpu().numpy()

    @torch.no_grad()
    def sample_action(self, x) -> np.ndarray:
        mu, sigma, _ = self(x)
        return self._dist(mu, sigma).sample().detach().cpu().numpy()


class DiagonalGMMPolicy(Policy):
    """Mixture of `num_components` squashed Normals per state (policies.py:502-669).  dist_head layout:
    [mu_0..mu_C-1 | log_std_0..log_std_C-1 | log_coeff_0..log_coeff_C-1] for action_dim == 1."""

    policy_type = "DiagonalGMM"

    def __init__(self, representation_dim: int, action_dim: int, action_bound: Optional[float], num_components: int,
                 hidden_dimensions: List[int], nonlinearity: str, layernorm: bool = False, log_param_min: float = -5,
                 log_param_max: float = 2):
        super().__init__()
        self._setup(representation_dim, action_dim, hidden_dimensions, nonlinearity, layernorm)
        if action_dim != 1:
            raise NotImplementedError("the engine implements one-dimensional actions")
        self.action_bound = action_bound
        self.log_param_min = log_param_min
        self.log_param_max = log_param_max
        self.num_components = num_components
        self.dist_head = nn.Linear(self.hidden_dimensions[-1], num_components * (2 * action_dim + 1))

    @property
    def bounds(self) -> np.ndarray:
        if self.action_bound is None:
            return np.array([-np.inf, np.inf], dtype=np.float32)
        return np.array([-self.action_bound, self.action_bound], dtype=np.float32)

    def forward(self, x):
        h = self.trunk(x)
        V_hat = self.value_head(h)
        params = self.dist_head(h)
        C_ = self.num_components
        dist_params = params[..., :C_ * 2 * self.action_dim].reshape(h.shape[0], -1)
        log_coeff = params[..., -C_:]
        mu, log_std = dist_params.chunk(2, dim=-1)
        log_std = torch.clamp(log_std, min=self.log_param_min, max=self.log_param_max)
        return mu, log_std.exp(), log_coeff, V_hat

    def _component(self, mu, sigma):
        if self.action_bound:
            return SquashedNormal(mu, sigma, self.action_bound)
        return torch.distributions.Normal(mu, sigma)

    def get_train_data(self, states, actions):
        """Mixture log-probs [B, K] = logsumexp_c(log_softmax(log_coeff)_c + log p_c(a)), entropy estimate [B], V_hat [B, 1]
        (policies.py:633-654; the squashed component's log|det J| sees x.shape[-1] == num_components there)."""
        mu, sigma, log_coeff, V_hat = self(states)
        K = actions.shape[-1]
        mu = mu.unsqueeze(dim=1).expand((-1, K, -1))
        sigma = sigma.unsqueeze(dim=1).expand((-1, K, -1))
        log_mix = torch.log_softmax(log_coeff, dim=-1).unsqueeze(dim=1).expand((-1, K, -1))
        comp_lp = self._component(mu, sigma).log_prob(actions.unsqueeze(-1))
        log_probs = torch.logsumexp(comp_lp + log_mix, dim=-1)
        return log_probs, -log_probs.mean(dim=-1), V_hat

    @torch.no_grad()
    def predict_V(self, x) -> np.ndarray:
        return self.value_head(self.trunk(x)).detach().cpu().numpy()

    @torch.no_grad()
    def sample_action(self, x) -> np.ndarray:
        mu, sigma, log_coeff, _ = self(x)
        comp = torch.distributions.Categorical(logits=log_coeff).sample().unsqueeze(-1)
        a = self._component(mu, sigma).sample()
        return torch.gather(a, -1, comp).detach().cpu().numpy()


_PolicyBase = Policy   # (earlier name)


def make_policy(representation_dim: int, action_dim: int, distribution: str, hidden_dimensions: List[int], nonlinearity: str,
                num_components: Optional[int] = None, num_actions: Optional[int] = None, action_bound: Optional[float] = None,
                layernorm: bool = False, log_param_min: float = -5, log_param_max: float = 2):
    """Factory with the reference's signature (policies.py:806-916)."""
    distribution = distribution.lower().replace(" ", "")
    if distribution == "discrete":
        return DiscretePolicy(representation_dim, action_dim, int(num_actions), hidden_dimensions, nonlinearity, layernorm)
    if distribution == "beta":
        raise NotImplementedError("GeneralizedBetaPolicy is not built (the reference marks it as not working, README.md:21-22)")
    assert num_components
    if num_components > 1:
        return DiagonalGMMPolicy(representation_dim, action_dim, action_bound, num_components, hidden_dimensions, nonlinearity, layernorm,
                                 log_param_min, log_param_max)
    return DiagonalNormalPolicy(representation_dim, action_dim, action_bound, hidden_dimensions, nonlinearity, layernorm,
                                log_param_min, log_param_max)
