"""Training losses (PyTorch autograd; the optimiser step is the one part of the stack that stays in PyTorch-ROCm).
Semantics follow alphazero/agent/losses.py: AlphaZeroLoss 30-151, A0CLoss 154-326, A0CLossTuned 329-500."""
from typing import Dict, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class Loss(nn.Module):
    """What every loss of this module is: a module whose forward returns a dict with at least "loss", "policy_loss" and
    "value_loss" (the reference's abstract base, losses.py:12-27)."""

    def forward(self, *args, **kwargs) -> Dict[str, torch.Tensor]:
        raise NotImplementedError


class AlphaZeroLoss(Loss):
    """policy_coeff * CE(logits, argmax of the MCTS policy) + value_coeff * MSE(V_hat, V)."""

    def __init__(self, policy_coeff: float, value_coeff: float, reduction: str) -> None:
        super().__init__()
        self.name = type(self).__name__
        self.policy_coeff, self.value_coeff, self.reduction = policy_coeff, value_coeff, reduction

    def forward(self, pi_prior_logits, pi_mcts, V_hat, V) -> Dict[str, torch.Tensor]:
        policy_loss = self.policy_coeff * F.cross_entropy(pi_prior_logits, pi_mcts.argmax(dim=1), reduction=self.reduction)
        value_loss = self.value_coeff * F.mse_loss(V_hat, V, reduction=self.reduction)
        return {"loss": policy_loss + value_loss, "policy_loss": policy_loss, "value_loss": value_loss}


class A0CLoss(Loss):
    """A0C loss: REINFORCE-style policy term sum_i (log pi_i - tau log n_i).detach() * log pi_i, entropy bonus, value MSE."""

    def __init__(self, tau: float, policy_coeff: float, alpha: Union[float, torch.Tensor], value_coeff: float, reduction: str) -> None:
        super().__init__()
        self.tau, self.policy_coeff, self.alpha, self.value_coeff, self.reduction = tau, policy_coeff, alpha, value_coeff, reduction

    def _reduce(self, x):
        return x.mean() if self.reduction == "mean" else x.sum()

    def _calculate_policy_loss(self, log_probs, counts):
        with torch.no_grad():
            log_diff = log_probs - self.tau * torch.log(counts)
        return self._reduce(torch.einsum("ni, ni -> n", log_diff, log_probs))

    def _calculate_value_loss(self, V_hat, V):
        return F.mse_loss(V_hat, V, reduction=self.reduction)

    def forward(self, log_probs, counts, entropy, V, V_hat) -> Dict[str, torch.Tensor]:
        policy_loss = self.policy_coeff * self._calculate_policy_loss(log_probs, counts)
        value_loss = self.value_coeff * self._calculate_value_loss(V_hat, V)
        entropy_loss = self.alpha * self._reduce(entropy)
        return {"loss": policy_loss + entropy_loss + value_loss, "policy_loss": policy_loss, "entropy_loss": entropy_loss,
                "value_loss": value_loss}


class A0CLossTuned(A0CLoss):
    """A0C loss whose entropy temperature alpha = exp(log_alpha) is itself trained (its own Adam step inside forward,
    losses.py:458-500) towards the target entropy -action_dim."""

    def __init__(self, action_dim: int, alpha_init: float, lr: float, tau: float, policy_coeff: float, value_coeff: float,
                 reduction: str, grad_clip: float, device: str) -> None:
        nn.Module.__init__(self)
        self.clip = grad_clip
        self.device = torch.device(device)
        self.target_entropy = -action_dim
        self.log_alpha = torch.tensor(np.log(alpha_init), requires_grad=True, device=self.device, dtype=torch.float32)
        alpha = self.log_alpha.exp()
        self.optimizer = torch.optim.Adam([self.log_alpha], lr=lr)
        self.tau, self.policy_coeff, self.alpha, self.value_coeff, self.reduction = tau, policy_coeff, alpha, value_coeff, reduction

    def _update_alpha(self, entropy):
        self.log_alpha.grad = None
        alpha_loss = (self.alpha * (entropy - self.target_entropy).detach()).mean()
        alpha_loss.backward()
        if self.clip:
            torch.nn.utils.clip_grad_norm_(self.log_alpha, self.clip)
        self.optimizer.step()
        self.alpha = self.log_alpha.exp()
        return alpha_loss.detach().cpu()

    def forward(self, log_probs, counts, entropy, V, V_hat) -> Dict[str, torch.Tensor]:
        policy_loss = self.policy_coeff * self._calculate_policy_loss(log_probs, counts)
        value_loss = self.value_coeff * self._calculate_value_loss(V_hat, V)
        entropy_loss = self.alpha.detach().item() * self._reduce(entropy)
        loss = policy_loss + entropy_loss + value_loss
        alpha_loss = self._update_alpha(entropy)
        return {"loss": loss, "policy_loss": policy_loss, "entropy_loss": entropy_loss, "value_loss": value_loss, "alpha_loss": alpha_loss}
