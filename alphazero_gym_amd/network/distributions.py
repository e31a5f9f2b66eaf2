"""Squashed Normal on (-c, c): a = c * tanh(x), x ~ N(mu, sigma)  (alphazero/network/distributions.py:10-109, 205-275).

Written directly (no torch.distributions.Transform machinery).  `log_prob` reproduces the reference's arithmetic,
including its quirk that the log|det J| term uses `x.shape[-1] * log(bound)` (distributions.py:107), i.e. the number of
actions evaluated per state rather than the action dimension.
"""
import math

import torch
import torch.nn.functional as F


class SquashedNormal:
    def __init__(self, loc: torch.Tensor, scale: torch.Tensor, bound: float, epsilon: float = 1e-6):
        assert bound > 0, "Scaling factor must be positive."
        self.loc, self.scale, self.bound, self.epsilon = loc, scale, float(bound), epsilon

    @property
    def mean(self) -> torch.Tensor:
        return self.bound * torch.tanh(self.loc)

    def sample(self, sample_shape=torch.Size()) -> torch.Tensor:
        with torch.no_grad():
            return self.rsample(sample_shape)

    def rsample(self, sample_shape=torch.Size()) -> torch.Tensor:
        shape = torch.Size(sample_shape) + self.loc.shape
        eps = torch.randn(shape, dtype=self.loc.dtype, device=self.loc.device)
        return self.bound * torch.tanh(self.loc + self.scale * eps)

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        x = torch.atanh(value / (self.bound + self.epsilon))
        var = self.scale ** 2
        base = -((x - self.loc) ** 2) / (2 * var) - torch.log(self.scale) - math.log(math.sqrt(2 * math.pi))
        corr = 1 + self.epsilon / self.bound
        ladj = x.shape[-1] * math.log(self.bound) + 2.0 * (math.log(2.0) - corr * x - F.softplus(-2.0 * corr * x))
        return base - ladj
