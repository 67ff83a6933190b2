"""`ExactMarginalLogLikelihood` with gpytorch's contract (experiments.py:233, README.md:45):
mll(model(X), Y) = likelihood(model(X)).log_prob(Y) / num_data, num_data =
function_dist.event_shape.numel() [gpytorch-knowledge, v1.11 -- the same expression the
reference copies at projected_lmc.py:1194].  `_add_other_terms` adds the model's added-loss terms and
the log-density of every registered lengthscale prior (priors.py)."""
import torch

from .distributions import MultivariateNormal
from .likelihoods import _GaussianLikelihoodBase


class MarginalLogLikelihood(torch.nn.Module):
    def __init__(self, likelihood, model):
        super().__init__()
        self.likelihood = likelihood
        self.model = model

    def _add_other_terms(self, res, params):
        """Added loss terms registered by the model's modules (the SGPR trace term of
        InducingPointKernel), then `prior.log_prob(value).sum()` of every registered prior, added to
        every element of `res` [gpytorch-knowledge: MarginalLogLikelihood._add_other_terms, v1.11]."""
        from .priors import named_priors
        for mod in self.model.modules():
            fn = getattr(mod, "added_loss_term", None)
            if fn is not None:
                term = fn()
                if term is not None:
                    res = res + term.reshape(res.shape)
        for _, _, prior, value in named_priors(self.model):
            res = res + prior.log_prob(value).sum()
        return res


class ExactMarginalLogLikelihood(MarginalLogLikelihood):
    def __init__(self, likelihood, model):
        if not isinstance(likelihood, _GaussianLikelihoodBase):
            raise RuntimeError("Likelihood must be Gaussian for exact inference")
        super().__init__(likelihood, model)

    def forward(self, function_dist, target, *params):
        if not isinstance(function_dist, MultivariateNormal):
            raise RuntimeError("ExactMarginalLogLikelihood can only operate on Gaussian random variables")
        output = self.likelihood(function_dist, *params)
        res = output.log_prob(target)
        res = self._add_other_terms(res, params)
        num_data = function_dist.event_shape.numel()
        return res / num_data
