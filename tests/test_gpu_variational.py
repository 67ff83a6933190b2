"""GPU parity of the variational LMC path (`VariationalMultitaskGPModel` + `VariationalELBO`,
SURVEY.md 8a row a12 / BASELINE config 4) against the CPU oracle: ELBO value and the gradient of
every parameter, including the learned inducing locations."""
import warnings

import pytest
import torch

from oracle import variational as ov
from oracle import gp_math as gm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def plmc():
    import projectedlmc
    assert torch.cuda.is_available()
    return projectedlmc


def _build(plmc, n, d, p, q, kernel, oscale, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = (2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1).to(dtype)
    Y = torch.randn(n, p, generator=g, dtype=torch.float64).to(dtype)
    torch.manual_seed(seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = plmc.VariationalMultitaskGPModel(X, n_latents=q, n_tasks=p, train_ind_ratio=1.5, seed=0,
                                                     init_lmc_coeffs=True, train_y=Y, mean_type=plmc.ConstantMean,
                                                     kernel_type=getattr(plmc, kernel), outputscales=oscale)
    finally:
        torch.set_default_dtype(old)
    g2 = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, prm in list(model.named_parameters()) + list(lik.named_parameters()):
            prm.add_(0.1 * torch.randn(prm.shape, generator=g2, dtype=torch.float64).to(prm.dtype))
    model.variational_strategy.base_variational_strategy.variational_params_initialized.fill_(1)
    return X, Y, model, lik


def _oracle_elbo(model, lik, X, Y, kind, nu, dtype_jitter):
    sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in
          list(model.named_parameters()) + [("lik." + k, v) for k, v in lik.named_parameters()]}
    b = "variational_strategy.base_variational_strategy."
    has_os = "covar_module.raw_outputscale" in sd
    ell = gm.softplus(sd["covar_module.base_kernel.raw_lengthscale" if has_os else "covar_module.raw_lengthscale"])
    ell = ell.reshape(ell.shape[0], -1)
    osc = gm.softplus(sd["covar_module.raw_outputscale"]) if has_os else None
    F = sd["lik.task_noise_covar_factor"]
    noise_diag = (F * F).sum(-1) + gm.softplus(sd["lik.raw_noise"]).reshape(()) + 1e-4
    val = ov.variational_elbo(kind, X.double(), Y.double(), sd[b + "inducing_points"], ell,
                              sd[b + "_variational_distribution.variational_mean"],
                              sd[b + "_variational_distribution.chol_variational_covar"],
                              sd["variational_strategy.lmc_coefficients"], noise_diag,
                              task_means=sd["variational_strategy.output_mean_module.raw_constant"],
                              nu=nu, outputscale=osc, jitter=dtype_jitter, num_data=X.shape[0])
    return val, sd


@pytest.mark.parametrize("kernel,oscale", [("RBFKernel", False), ("MaternKernel", True)])
def test_elbo_and_all_gradients_fp64(plmc, kernel, oscale):
    n, d, p, q = 150, 3, 4, 2
    X, Y, model, lik = _build(plmc, n, d, p, q, kernel, oscale, torch.float64)
    kind, nu = ("rbf", 2.5) if kernel == "RBFKernel" else ("matern", 2.5)
    ref, sd = _oracle_elbo(model, lik, X, Y, kind, nu, 1e-6)
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    mll = plmc.VariationalELBO(lik, model, num_data=n)
    out = mll(model(X.to(DEV)), Y.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-9 * abs(float(ref)), (float(out), float(ref))
    named = dict(list(model.named_parameters()) + [("lik." + k, v) for k, v in lik.named_parameters()])
    for name, leaf in sd.items():
        if leaf.grad is None:                       # unused by the ELBO (e.g. raw_task_noises does not exist for rank>0)
            continue
        got = named[name].grad
        assert got is not None, name
        ref_g = leaf.grad
        if name.endswith("chol_variational_covar"):
            got, ref_g = got.cpu().tril(), ref_g.tril()
        assert torch.allclose(got.cpu(), ref_g, rtol=2e-5, atol=1e-8), (name, (got.cpu() - ref_g).abs().max())
    assert named["variational_strategy.base_variational_strategy.inducing_points"].grad.abs().max() > 0


def test_elbo_fp32_config4_shape(plmc):
    """Scaled-down BASELINE config 4 (16 tasks, q = 8, Cholesky variational distribution), fp32:
    ELBO within 1e-4 relative of the fp64 oracle evaluated at the same (fp32-rounded) parameters."""
    n, d, p, q = 600, 8, 16, 8
    X, Y, model, lik = _build(plmc, n, d, p, q, "RBFKernel", False, torch.float32, seed=3)
    ref, _ = _oracle_elbo(model, lik, X, Y, "rbf", 2.5, 1e-4)
    model, lik = model.to(DEV), lik.to(DEV)
    out = plmc.VariationalELBO(lik, model, num_data=n)(model(X.to(DEV)), Y.to(DEV))
    assert abs(float(out) - float(ref)) < 1e-4 * abs(float(ref)), (float(out), float(ref))
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model(X[:50].to(DEV)))
    assert pred.mean.shape == (50, p) and bool((pred.variance > 0).all())


# ------------------------------------------------------------------------------------------------
# train_ind_ratio == 1: UnwhitenedVariationalStrategy with inducing points = training inputs
# (projected_lmc.py:724-729)
def _build_unwhitened(plmc, n, d, p, q, kernel, oscale, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(n, p, generator=g, dtype=torch.float64)
    torch.manual_seed(seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=2)
        with pytest.warns(UserWarning, match="inducing points not learned"):
            model = plmc.VariationalMultitaskGPModel(X, n_latents=q, n_tasks=p, train_ind_ratio=1.0, seed=0,
                                                     init_lmc_coeffs=True, train_y=Y, mean_type=plmc.ConstantMean,
                                                     kernel_type=getattr(plmc, kernel), outputscales=oscale)
    finally:
        torch.set_default_dtype(old)
    return X, Y, model, lik


def _perturb(model, lik, seed):
    g2 = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, prm in list(model.named_parameters()) + list(lik.named_parameters()):
            prm.add_(0.1 * torch.randn(prm.shape, generator=g2, dtype=torch.float64).to(prm.dtype))


def _oracle_elbo_unwhitened(model, lik, X, Y, kind, nu, Xeval=None):
    sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in
          list(model.named_parameters()) + [("lik." + k, v) for k, v in lik.named_parameters()]}
    b = "variational_strategy.base_variational_strategy."
    has_os = "covar_module.raw_outputscale" in sd
    ell = gm.softplus(sd["covar_module.base_kernel.raw_lengthscale" if has_os else "covar_module.raw_lengthscale"])
    ell = ell.reshape(ell.shape[0], -1)
    osc = gm.softplus(sd["covar_module.raw_outputscale"]) if has_os else None
    F = sd["lik.task_noise_covar_factor"]
    noise_diag = (F * F).sum(-1) + gm.softplus(sd["lik.raw_noise"]).reshape(()) + 1e-4
    Z = model.variational_strategy.base_variational_strategy.inducing_points.detach().cpu().double()
    val = ov.variational_elbo(kind, X.double(), Y.double(), Z, ell,
                              sd[b + "_variational_distribution.variational_mean"],
                              sd[b + "_variational_distribution.chol_variational_covar"],
                              sd["variational_strategy.lmc_coefficients"], noise_diag,
                              task_means=sd["variational_strategy.output_mean_module.raw_constant"],
                              nu=nu, outputscale=osc, jitter=1e-3, num_data=X.shape[0], whitened=False)
    return val, sd, (ell, osc, Z)


@pytest.mark.parametrize("kernel,oscale", [("RBFKernel", False), ("MaternKernel", True)])
def test_unwhitened_elbo_and_all_gradients_fp64(plmc, kernel, oscale):
    n, d, p, q = 140, 2, 3, 2
    X, Y, model, lik = _build_unwhitened(plmc, n, d, p, q, kernel, oscale)
    kind, nu = ("rbf", 2.5) if kernel == "RBFKernel" else ("matern", 2.5)
    bvs = model.variational_strategy.base_variational_strategy
    assert type(bvs).__name__ == "UnwhitenedVariationalStrategy"
    assert "inducing_points" not in dict(bvs.named_parameters())            # not learned (:727)
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    mll = plmc.VariationalELBO(lik, model, num_data=n)
    # first call initialises q(u) to the prior: chol_variational_covar <- chol(K_ZZ + 1e-3 I)
    with torch.no_grad():
        mll(model(X.to(DEV)), Y.to(DEV))
    ell0 = model.covar_module.base_kernel.lengthscale if oscale else model.covar_module.lengthscale
    osc0 = model.covar_module.outputscale.detach().cpu().double() if oscale else None
    K0 = gm.kernel_matrix(kind, X, X, ell0.detach().cpu().double().reshape(q, d), osc0, nu) + 1e-3 * torch.eye(n, dtype=torch.float64)
    Ls0 = bvs._variational_distribution.chol_variational_covar.detach().cpu()
    assert torch.allclose(Ls0 @ Ls0.transpose(-1, -2), K0, rtol=1e-9, atol=1e-11)
    # now an arbitrary point of parameter space
    model, lik = model.cpu(), lik.cpu()
    _perturb(model, lik, 5)
    ref, sd, _ = _oracle_elbo_unwhitened(model, lik, X, Y, kind, nu)
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    out = mll(model(X.to(DEV)), Y.to(DEV))
    out.backward()
    assert abs(float(out.detach()) - float(ref.detach())) < 1e-9 * abs(float(ref.detach())), (float(out), float(ref))
    named = dict(list(model.named_parameters()) + [("lik." + k, v) for k, v in lik.named_parameters()])
    for name, leaf in sd.items():
        if leaf.grad is None:
            continue
        got, ref_g = named[name].grad, leaf.grad
        assert got is not None, name
        if name.endswith("chol_variational_covar"):
            got, ref_g = got.cpu().tril(), ref_g.tril()
        assert torch.allclose(got.cpu(), ref_g, rtol=2e-5, atol=1e-8), (name, (got.cpu() - ref_g).abs().max())


def test_unwhitened_eval_predictions_fp64(plmc):
    n, d, p, q, ns = 120, 2, 3, 2, 17
    X, Y, model, lik = _build_unwhitened(plmc, n, d, p, q, "MaternKernel", False, seed=2)
    model.variational_strategy.base_variational_strategy.variational_params_initialized.fill_(1)
    _perturb(model, lik, 9)
    g = torch.Generator().manual_seed(11)
    Xs = 2 * torch.rand(ns, d, generator=g, dtype=torch.float64) - 1
    _, sd, (ell, osc, Z) = _oracle_elbo_unwhitened(model, lik, X, Y, "matern", 2.5)
    b = "variational_strategy.base_variational_strategy."
    with torch.no_grad():
        mean_f, var_f, _ = ov.unwhitened_latent_predictive("matern", Xs, Z, ell, sd[b + "_variational_distribution.variational_mean"],
                                                           sd[b + "_variational_distribution.chol_variational_covar"], 2.5, osc, 1e-3)
        H = sd["variational_strategy.lmc_coefficients"]
        mu_ref = mean_f.T @ H + sd["variational_strategy.output_mean_module.raw_constant"].reshape(1, p)
        var_ref = var_f.T @ (H * H)
    model = model.to(DEV)
    model.eval()
    with torch.no_grad():
        dist = model(Xs.to(DEV))
    assert torch.allclose(dist.mean.cpu(), mu_ref, rtol=1e-8, atol=1e-10)
    assert torch.allclose(dist.variance.cpu(), var_ref, rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_triangular_gemm_ranges_give_the_dense_product(dtype, tol):
    """plmc_gemm_tn_tri (the products of the Cholesky adjoint behind VariationalMultitaskGPModel, projected_lmc.py:672-683): with
    triangular operands declared, a tile only walks the contraction range that has entries -- the result must be the dense
    product.  Sizes that are not block multiples (m = 300, n = 200: padded by the host), every combination the adjoint uses."""
    from projectedlmc._dense import gemm_tn, TRI_A_LOWER as AL, TRI_B_LOWER as BL, TRI_A_UPPER as AU, TRI_C_LOWER as CL, TRI_C_ZERO as CZ
    q, m, n = 2, 300, 200
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(DEV, dtype)
    Lo, Lo2, Up, F = torch.tril(rnd(q, m, m)), torch.tril(rnd(q, m, m)), torch.triu(rnd(q, m, m)), rnd(q, m, n)
    mT = lambda t: t.transpose(-1, -2)
    ref = lambda A, B: mT(A).double() @ B.double()
    def close(got, want):
        assert (got.double() - want).abs().max() < tol * max(1.0, float(want.abs().max())), float((got.double() - want).abs().max())
    close(gemm_tn(Lo, F, AL), ref(Lo, F))                                              # W^T G
    close(torch.tril(gemm_tn(mT(F), mT(F[:, :, :].contiguous()), CL)), torch.tril(ref(mT(F), mT(F))))   # tril(C A^T)
    close(torch.tril(gemm_tn(mT(Up), Lo, AL | BL | CL)), torch.tril(ref(mT(Up), Lo)))  # tril(U Lbar): U^T stored K-major is lower
    PW = gemm_tn(mT(Lo), Lo2, AU | BL | CL | CZ)                                       # P W (lower times lower), zeros above
    close(PW, ref(mT(Lo), Lo2))
    close(gemm_tn(Lo2, PW, AL | BL), ref(Lo2, ref(mT(Lo), Lo2).to(dtype)))             # W^T (P W)
    close(gemm_tn(mT(Lo), F, AU), ref(mT(Lo), F))                                      # Ls G


def test_lower_t_matmul_matches_torch_autograd():
    from projectedlmc import _var_engine
    q, m, n = 2, 260, 150
    g = torch.Generator().manual_seed(4)
    Ls0 = torch.tril(torch.randn(q, m, m, generator=g, dtype=torch.float64)).to(DEV)
    A0 = torch.randn(q, m, n, generator=g, dtype=torch.float64).to(DEV)
    Gout = torch.randn(q, m, n, generator=g, dtype=torch.float64).to(DEV)
    Ls1, A1 = Ls0.clone().requires_grad_(), A0.clone().requires_grad_()
    (_var_engine.lower_t_matmul(Ls1.tril(), A1) * Gout).sum().backward()
    Ls2, A2 = Ls0.clone().requires_grad_(), A0.clone().requires_grad_()
    ((Ls2.tril().transpose(-1, -2) @ A2) * Gout).sum().backward()
    assert torch.allclose(Ls1.grad, Ls2.grad, rtol=1e-10, atol=1e-10) and torch.allclose(A1.grad, A2.grad, rtol=1e-10, atol=1e-10)
