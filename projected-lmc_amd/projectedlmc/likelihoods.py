"""Gaussian likelihoods with the gpytorch parameter names and semantics the reference uses:
`GaussianLikelihood(batch_shape=[q], noise_constraint=GreaterThan(e^noise_thresh))`
(projected_lmc.py:920-921) and `MultitaskGaussianLikelihood(num_tasks, rank, has_global_noise)`
(:1025, experiments.py:184,190).  Applying a likelihood to a prior only records the noise on the
lazy covariance; the addition itself is fused into the HIP assembly kernel.

[gpytorch-knowledge]: noise = softplus(raw_noise) + lower bound (default 1e-4), raw init 0;
multitask rank 0: Sigma = diag(task_noises) + noise I; rank r > 0: Sigma = F F^T + noise I with
F = task_noise_covar_factor (p x r, randn init); has_global_noise=False drops noise I.
"""
import torch

from .constraints import GreaterThan
from .distributions import MultivariateNormal, MultitaskMultivariateNormal, KroneckerSumCovariance
from .kernels import LazyKernel


class Likelihood(torch.nn.Module):
    pass


class _GaussianLikelihoodBase(Likelihood):
    pass


class HomoskedasticNoise(torch.nn.Module):
    def __init__(self, noise_constraint=None, batch_shape=torch.Size()):
        super().__init__()
        self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(*batch_shape, 1)))
        self.raw_noise_constraint = noise_constraint or GreaterThan(1e-4)

    @property
    def noise(self):
        return self.raw_noise_constraint.transform(self.raw_noise)

    @noise.setter
    def noise(self, value):
        value = torch.as_tensor(value, dtype=self.raw_noise.dtype, device=self.raw_noise.device)
        with torch.no_grad():
            self.raw_noise.copy_(self.raw_noise_constraint.inverse_transform(value).expand_as(self.raw_noise))


class GaussianLikelihood(_GaussianLikelihoodBase):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = torch.Size(batch_shape)
        self.noise_covar = HomoskedasticNoise(noise_constraint, self.batch_shape)

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.noise = value

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise

    def forward(self, function_dist, *params, **kwargs):
        c = function_dist.lazy_covariance_matrix
        noise = self.noise.reshape(-1)
        if isinstance(c, LazyKernel) or hasattr(c, "log_prob_batch"):
            new = c.add_noise(noise.to(c.ell.dtype))
        elif torch.is_tensor(c):
            eye = torch.eye(c.shape[-1], dtype=c.dtype, device=c.device)
            new = c + noise.reshape(*self.batch_shape, 1, 1) * eye
        else:
            raise NotImplementedError("GaussianLikelihood on %s" % type(c).__name__)
        return MultivariateNormal(function_dist.mean, new)


class MultitaskGaussianLikelihood(_GaussianLikelihoodBase):
    def __init__(self, num_tasks, rank=0, task_prior=None, batch_shape=torch.Size(), noise_prior=None,
                 noise_constraint=None, has_global_noise=True, has_task_noise=True, **kwargs):
        super().__init__()
        self.num_tasks, self.rank = num_tasks, rank
        self.has_global_noise, self.has_task_noise = has_global_noise, has_task_noise
        self.raw_noise_constraint = noise_constraint or GreaterThan(1e-4)
        if has_task_noise:
            if rank == 0:
                self.register_parameter("raw_task_noises", torch.nn.Parameter(torch.zeros(num_tasks)))
                self.raw_task_noises_constraint = noise_constraint or GreaterThan(1e-4)
            else:
                self.register_parameter("task_noise_covar_factor", torch.nn.Parameter(torch.randn(num_tasks, rank)))
        if has_global_noise:
            self.register_parameter("raw_noise", torch.nn.Parameter(torch.zeros(1)))

    @property
    def noise(self):
        return self.raw_noise_constraint.transform(self.raw_noise)

    @property
    def task_noises(self):
        if self.rank > 0:
            raise AttributeError("Cannot get diagonal task noises when covariance is rank %d" % self.rank)
        return self.raw_task_noises_constraint.transform(self.raw_task_noises)

    @property
    def task_noise_covar(self):
        if self.rank == 0:
            return torch.diag_embed(self.task_noises)
        return self.task_noise_covar_factor @ self.task_noise_covar_factor.T

    def __getattr__(self, name):
        # `hasattr(lik, "noise")` must be False without global noise (experiments.py:323,333)
        if name == "noise" and not self.__dict__.get("has_global_noise", True):
            raise AttributeError(name)
        return super().__getattr__(name)

    def task_noise_matrix(self, dtype=None):
        """Sigma (p x p)."""
        S = 0.0
        if self.has_task_noise:
            S = self.task_noise_covar
        if self.has_global_noise:
            ref = self.raw_noise
            S = S + self.noise.reshape(()) * torch.eye(self.num_tasks, dtype=ref.dtype, device=ref.device)
        return S if dtype is None else S.to(dtype)

    def expected_log_prob(self, target, input, *params, **kwargs):
        """E_q(f)[log p(y|f)] per data point; like gpytorch's _GaussianLikelihoodBase it uses only the
        marginal variances of q(f) and the DIAGONAL of the task-noise covariance [gpytorch-knowledge]."""
        import math
        mean, variance = input.mean, input.variance
        noise = torch.diagonal(self.task_noise_matrix(mean.dtype)).reshape(1, -1)
        res = ((target - mean).square() + variance) / noise + noise.log() + math.log(2 * math.pi)
        return res.mul(-0.5).sum(-1)

    def forward(self, function_dist, *params, **kwargs):
        c = function_dist.lazy_covariance_matrix
        Sigma = self.task_noise_matrix(function_dist.mean.dtype)
        if hasattr(c, "add_task_noise"):
            return MultitaskMultivariateNormal(function_dist.mean, c.add_task_noise(Sigma))
        ind = getattr(function_dist, "_independent", None)
        if ind is not None and self.rank == 0:
            noisy = ind.lazy_covariance_matrix.add_noise(torch.diagonal(Sigma).to(function_dist.mean.dtype))
            return MultitaskMultivariateNormal.from_batch_mvn(MultivariateNormal(ind.mean, noisy))
        raise NotImplementedError("MultitaskGaussianLikelihood on %s" % type(c).__name__)
