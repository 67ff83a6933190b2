"""Lengthscale priors (projected_lmc.py:135-149,169-179) and `kernel_cond` (:367):
the exact MLL of a model built with `prior_scales` equals the oracle MLL + sum(log prior) / n, and its
gradient follows."""
import pytest
import torch

from oracle import gp_math as gm
from oracle import priors as opr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _data(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    y = torch.sin(2.0 * X.sum(-1)) + 0.1 * torch.randn(n, generator=g, dtype=torch.float64)
    return X, y


@pytest.mark.parametrize("d", [1, 3])
def test_exact_mll_with_lengthscale_prior(d):
    import projectedlmc as plmc
    n = 280
    X, y = _data(n, d, 3)
    ps = torch.linspace(0.4, 0.9, d, dtype=torch.float64)
    pw = torch.linspace(0.3, 0.5, d, dtype=torch.float64)
    lik = plmc.GaussianLikelihood().double()
    model = plmc.ExactGPModel(X, y, lik, prior_scales=ps, prior_width=pw, mean_type=plmc.ZeroMean,
                              kernel_type=plmc.MaternKernel).double()
    # lengthscales start at the prior mean (:169-179)
    assert torch.allclose(model.covar_module.lengthscale.detach().reshape(-1), ps)
    with torch.no_grad():
        model.covar_module.raw_lengthscale.add_(0.3)
    ell = model.covar_module.lengthscale.detach().reshape(1, d).clone().requires_grad_(True)
    noise = lik.noise.detach().reshape(1)
    K = gm.kernel_matrix("matern", X, X, ell, None, 2.5) + noise.reshape(-1, 1, 1) * torch.eye(n, dtype=torch.float64)
    lp_ref = torch.distributions.MultivariateNormal(torch.zeros(n, dtype=torch.float64), covariance_matrix=K[0]).log_prob(y)
    ref = (lp_ref + opr.lengthscale_log_prior(ell, ps, pw)) / n
    (g_ref,) = torch.autograd.grad(ref, ell)

    model = model.to(DEV)
    lik = lik.to(DEV)
    model.train(); lik.train()
    mll = plmc.ExactMarginalLogLikelihood(lik, model)
    out = mll(model(X.to(DEV)), y.to(DEV))
    assert abs(float(out.detach()) - float(ref.detach())) <= 1e-9 * abs(float(ref.detach()))
    out.backward()
    # chain rule through softplus: d/d raw = d/d ell * sigmoid(raw)
    raw = model.covar_module.raw_lengthscale.detach().cpu().reshape(1, d)
    g = model.covar_module.raw_lengthscale.grad.cpu().reshape(1, d)
    assert torch.allclose(g, g_ref * torch.sigmoid(raw), rtol=2e-6, atol=1e-10)


def test_prior_requires_width():
    import projectedlmc as plmc
    with pytest.raises(ValueError, match="prior width"):
        plmc.handle_covar_(plmc.RBFKernel, 2, prior_scales=torch.ones(2))


def test_kernel_cond():
    import projectedlmc as plmc
    n, d = 200, 2
    X, y = _data(n, d, 5)
    lik = plmc.GaussianLikelihood().double()
    model = plmc.ExactGPModel(X, y, lik, mean_type=plmc.ZeroMean, kernel_type=plmc.RBFKernel).double()
    ell = model.covar_module.lengthscale.detach().reshape(1, d)
    K = gm.kernel_matrix("rbf", X, X, ell, None, 2.5)[0] + lik.noise.detach().reshape(()) * torch.eye(n, dtype=torch.float64)
    ref = torch.linalg.cond(K)
    model = model.to(DEV)
    model.likelihood = model.likelihood.to(DEV)
    got = model.kernel_cond()
    assert abs(float(got) - float(ref)) <= 1e-7 * float(ref)


@pytest.mark.parametrize("mean", ["LinearMean", "PolynomialMean"])
def test_reference_mean_functions(mean):
    """LinearMean / PolynomialMean of the reference (projected_lmc.py:38-81) on the exact-GP path: MLL and the
    gradient of every mean / kernel / noise parameter against the dense oracle; basis_matrix of LinearMean."""
    import projectedlmc as plmc
    from oracle import gp_math as gm
    g = torch.Generator().manual_seed(0)
    n, d = 140, 3
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    y = torch.randn(n, generator=g, dtype=torch.float64)
    torch.manual_seed(3)
    lik = plmc.GaussianLikelihood()
    model = plmc.ExactGPModel(X, y, lik, mean_type=getattr(plmc, mean), kernel_type=plmc.MaternKernel).double()
    lik = lik.double()
    leaves = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    if mean == "LinearMean":
        m = (X @ leaves["mean_module.weights"].reshape(d, 1)).reshape(-1) + leaves["mean_module.bias"].reshape(())
        assert torch.equal(model.mean_module.basis_matrix(X), torch.hstack([X, torch.ones(n, 1, dtype=torch.float64)]))
    else:
        m = sum(((X ** i) @ leaves["mean_module.weights_%d" % i].reshape(d, 1)).reshape(-1) for i in (1, 2, 3))
        m = m + leaves["mean_module.bias"].reshape(())
    ell = gm.softplus(leaves["covar_module.raw_lengthscale"]).reshape(1, -1)
    noise = gm.softplus(leaves["likelihood.noise_covar.raw_noise"]).reshape(1) + 1e-4
    ref = gm.exact_latent_log_prob("matern", X, ell, noise, (y - m).reshape(1, -1), None, 2.5).sum() / n
    ref.backward()
    model, lik = model.to("cuda:0"), lik.to("cuda:0")
    model.train(); lik.train()
    out = plmc.ExactMarginalLogLikelihood(lik, model)(model(X.to("cuda:0")), y.to("cuda:0")).sum()
    out.backward()
    assert abs(float(out.detach()) - float(ref)) < 1e-9 * abs(float(ref))
    for name, prm in model.named_parameters():
        if leaves[name].grad is None:                       # weights_0 of PolynomialMean is never used (as in the reference)
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0
            continue
        assert torch.allclose(prm.grad.cpu(), leaves[name].grad, rtol=1e-5, atol=1e-9), name
