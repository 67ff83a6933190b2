"""GPU parity of the inducing-point (SGPR) variant, `n_inducing_points` (SURVEY.md 8f row 2;
projected_lmc.py:302-303): MLL incl. the added trace term, gradients (lengthscales, noise, inducing
locations), predictions -- for ExactGPModel and for ProjectedGPModel latents."""
import warnings

import pytest
import torch

from oracle import sgpr as osg
from oracle import gp_math as gm
from oracle import projected as pj
from _bridge import oracle_params

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _data(n, d, p, seed):
    g = torch.Generator().manual_seed(seed)
    return (2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1, torch.randn(n, p, generator=g, dtype=torch.float64))


def test_exact_gp_sgpr_mll_grads_and_prediction():
    import projectedlmc as plmc
    n, d, m = 220, 2, 30
    X, Y = _data(n, d, 1, 1)
    y = Y[:, 0].contiguous()
    torch.manual_seed(4)
    lik = plmc.GaussianLikelihood()
    model = plmc.ExactGPModel(X, y, lik, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel, n_inducing_points=m)
    model, lik = model.double(), lik.double()
    with torch.no_grad():
        model.covar_module.inducing_points.copy_(2 * torch.rand(m, d, dtype=torch.float64) - 1)
        model.covar_module.base_kernel.raw_lengthscale.fill_(-1.0)        # ell = 0.31: well-conditioned K_zz
    Z = model.covar_module.inducing_points.detach().clone().requires_grad_()
    raw_ls = model.covar_module.base_kernel.raw_lengthscale.detach().clone().requires_grad_()
    raw_nz = lik.noise_covar.raw_noise.detach().clone().requires_grad_()
    ell, nz = gm.softplus(raw_ls).reshape(1, d), gm.softplus(raw_nz).reshape(1) + 1e-4
    lp, tr = osg.sgpr_terms("matern", X, Z, ell, nz, y[None])
    ref = (lp + tr).sum() / n
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    out = plmc.ExactMarginalLogLikelihood(lik, model)(model(X.to(DEV)), y.to(DEV)).sum()
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-8 * abs(float(ref)), (float(out), float(ref))
    assert torch.allclose(model.covar_module.inducing_points.grad.cpu(), Z.grad, rtol=1e-5, atol=1e-8)
    assert torch.allclose(model.covar_module.base_kernel.raw_lengthscale.grad.cpu(), raw_ls.grad, rtol=1e-5, atol=1e-8)
    assert torch.allclose(lik.noise_covar.raw_noise.grad.cpu(), raw_nz.grad, rtol=1e-5, atol=1e-8)
    Xs = 2 * torch.rand(25, d, dtype=torch.float64) - 1
    mu, cov = osg.sgpr_posterior("matern", X, Z.detach(), ell.detach(), nz.detach(), y[None], Xs)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = model(Xs.to(DEV))
    assert torch.allclose(pred.mean.cpu(), mu[0], rtol=1e-6, atol=1e-8)
    assert torch.allclose(pred.variance.cpu(), torch.diagonal(cov[0]), rtol=1e-5, atol=1e-8)


def test_projected_model_with_inducing_points():
    import projectedlmc as plmc
    n, d, p, q, m = 160, 2, 4, 2, 25
    X, Y = _data(n, d, p, 2)
    torch.manual_seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = plmc.ProjectedGPModel(X, Y, p, q, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                      init_lmc_coeffs=True, BDN=True, scalar_B=True, diagonal_B=True,
                                      n_inducing_points=m).double()
    with torch.no_grad():
        model.covar_module.inducing_points.copy_(2 * torch.rand(m, d, dtype=torch.float64) - 1)
        model.covar_module.base_kernel.raw_lengthscale.fill_(-1.0)
    sd = model.state_dict()
    P = dict(oracle_params_like(model, sd))
    ytil = pj.project_data(P, Y)
    lp, tr = osg.sgpr_terms("matern", X, sd["covar_module.inducing_points"].double(), pj.lengthscale(P),
                            pj.projected_noise(P), ytil)
    terms, const = pj.projection_terms(P, Y)
    ref = float((lp + tr).sum() / n + sum(terms) + const)
    model = model.to(DEV)
    model.train()
    val = float(plmc.ProjectedLMCmll(model.likelihood, model)(model(X.to(DEV)), Y.to(DEV)))
    assert abs(val - ref) < 1e-8 * abs(ref), (val, ref)
    model.eval()
    with torch.no_grad():
        pred = model(X[:10].to(DEV))
        full = model(X[:10].to(DEV), full_cov=True)                 # inducing-point latents with the full task covariance (:1149-1152)
    assert pred.mean.shape == (10, p) and bool((pred.variance > 0).all())
    # oracle: latent SGPR posteriors mixed as :1143-1153
    Z = sd["covar_module.inducing_points"].double().cpu()
    mu_lat, cov_lat = osg.sgpr_posterior("matern", X, Z, pj.lengthscale(P), pj.projected_noise(P), ytil, X[:10])
    Ht = pj.lmc_coefficients(P)
    mean_ref = mu_lat.T @ Ht
    cov_ref = sum(torch.kron(cov_lat[i], torch.outer(Ht[i], Ht[i])) for i in range(q)) + P["eps"] * torch.eye(10 * p, dtype=torch.float64)
    assert torch.allclose(pred.mean.cpu(), mean_ref, rtol=1e-6, atol=1e-8)
    assert torch.allclose(pred.variance.cpu(), torch.diagonal(cov_ref).reshape(10, p), rtol=1e-5, atol=1e-8)
    assert torch.allclose(full.mean.cpu(), mean_ref, rtol=1e-6, atol=1e-8)
    assert torch.allclose(full.lazy_covariance_matrix.evaluate().cpu(), cov_ref, rtol=1e-5, atol=1e-8)


def oracle_params_like(model, sd):
    """oracle_params for a model whose kernel is wrapped by InducingPointKernel."""
    import math
    lb = model.likelihood.noise_covar.raw_noise_constraint.lower_bound
    P = dict(kind="matern", nu=2.5, n_tasks=model.n_tasks, n_latents=model.n_latents, mode=model.lmc_coefficients.mode,
             BDN=True, eps=model.eps, scalar_B=True, diagonal_B=True, noise_lb=lb, noise_thresh=math.log(lb),
             H=sd["lmc_coefficients.H"].cpu().double(), raw_noise=sd["likelihood.noise_covar.raw_noise"].cpu().double(),
             raw_lengthscale=sd["covar_module.base_kernel.raw_lengthscale"].cpu().double(), raw_outputscale=None,
             log_B_tilde=sd["parametrizations.log_B_tilde.original"].cpu().double())
    return P
