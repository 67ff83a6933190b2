"""Lengthscale priors the reference's kernel factory registers (projected_lmc.py:135-149):
`gp.priors.NormalPrior(loc, scale)` for one-variable kernels and
`gp.priors.MultivariateNormalPrior(loc, covariance_matrix)` for ARD groups.  gpytorch semantics
[gpytorch-knowledge, v1.11]: a prior is a torch.distributions object that is also a Module (its
tensors follow `.to()` / `.cuda()`); marginal log-likelihoods add
`prior.log_prob(closure(module)).sum()` for every registered prior (`_add_other_terms`), the closure
of a `lengthscale_prior` being `module.lengthscale`."""
import math

import torch


class Prior(torch.nn.Module):
    def log_prob(self, x):
        raise NotImplementedError

    @property
    def mean(self):
        return self.loc


class NormalPrior(Prior):
    def __init__(self, loc, scale, validate_args=False, transform=None):
        super().__init__()
        self.register_buffer("loc", torch.as_tensor(loc, dtype=torch.get_default_dtype()).clone())
        self.register_buffer("scale", torch.as_tensor(scale, dtype=torch.get_default_dtype()).clone())

    @property
    def variance(self):
        return self.scale ** 2

    def log_prob(self, x):
        loc, scale = self.loc.to(x.dtype), self.scale.to(x.dtype)
        return -0.5 * ((x - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2 * math.pi)


class MultivariateNormalPrior(Prior):
    """Event = the last dimension (the ARD lengthscales of one kernel)."""

    def __init__(self, loc, covariance_matrix=None, precision_matrix=None, scale_tril=None, validate_args=False,
                 transform=None):
        super().__init__()
        if covariance_matrix is None:
            if scale_tril is not None:
                covariance_matrix = scale_tril @ scale_tril.transpose(-1, -2)
            elif precision_matrix is not None:
                covariance_matrix = torch.linalg.inv(precision_matrix)
            else:
                raise ValueError("one of covariance_matrix, precision_matrix, scale_tril is needed")
        self.register_buffer("loc", torch.as_tensor(loc, dtype=torch.get_default_dtype()).clone())
        self.register_buffer("covariance_matrix", torch.as_tensor(covariance_matrix, dtype=torch.get_default_dtype()).clone())

    def log_prob(self, x):
        loc, cov = self.loc.to(x.dtype), self.covariance_matrix.to(x.dtype)
        L = torch.linalg.cholesky(cov)
        diff = (x - loc).unsqueeze(-1)
        z = torch.linalg.solve_triangular(L, diff, upper=False).squeeze(-1)
        half_logdet = torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
        d = x.shape[-1]
        return -0.5 * (z ** 2).sum(-1) - half_logdet - 0.5 * d * math.log(2 * math.pi)


def named_priors(module):
    """(name, module, prior, value) for every lengthscale prior registered below `module`."""
    out = []
    for name, mod in module.named_modules():
        pr = getattr(mod, "lengthscale_prior", None)
        if isinstance(pr, Prior):
            out.append((name + ".lengthscale_prior", mod, pr, mod.lengthscale))
    return out
