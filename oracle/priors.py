"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the lengthscale priors of projected_lmc.py:135-149.
PARITY UNPINNED (no reference fixtures; gpytorch is not installable here).

gpytorch 1.11 semantics restated [gpytorch-knowledge]: NormalPrior(loc, scale) and
MultivariateNormalPrior(loc, covariance_matrix) are the torch.distributions densities of those
names; a marginal log-likelihood adds sum(prior.log_prob(lengthscale)) to every element of its
batched result before dividing by the number of data (MarginalLogLikelihood._add_other_terms)."""
import math

import torch


def normal_logpdf(x, loc, scale):
    return -0.5 * ((x - loc) / scale) ** 2 - torch.log(scale) - 0.5 * math.log(2.0 * math.pi)


def mvn_diag_logpdf(x, loc, var_diag):
    """log N(x; loc, diag(var_diag)), event = last dimension."""
    d = x.shape[-1]
    return (-0.5 * ((x - loc) ** 2 / var_diag).sum(-1) - 0.5 * torch.log(var_diag).sum() - 0.5 * d * math.log(2.0 * math.pi))


def lengthscale_log_prior(ell, prior_scales, prior_width):
    """Total log prior of ARD lengthscales `ell` (..., d) under the reference's construction for ONE kernel over
    all d variables (:140-145): MVN with covariance diag(prior_scales * prior_width) when d > 1, else
    Normal with deviation prior_scales * prior_width."""
    d = ell.shape[-1]
    if d > 1:
        return mvn_diag_logpdf(ell, prior_scales, prior_scales * prior_width).sum()
    return normal_logpdf(ell, prior_scales, prior_scales * prior_width).sum()
