"""GPU parity of the exact dense LMC / ICM path (`MultitaskGPModel`, SURVEY.md 8a row a8 / BASELINE
config 2) against the CPU oracle: MLL, gradients of every parameter, predictions."""
import pytest
import torch

from oracle import lmc_dense as ld
from oracle import gp_math as gm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def plmc():
    import projectedlmc
    assert torch.cuda.is_available()
    return projectedlmc


def _data(n, d, p, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1,
            torch.randn(n, p, generator=g, dtype=torch.float64))


def _oracle_inputs(model, lik):
    """Raw parameters of the product model as leaf tensors + the oracle's constrained quantities."""
    sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in
          [kv for kv in model.named_parameters() if not kv[0].startswith("likelihood.")]
          + [("lik." + k, v) for k, v in lik.named_parameters()]}
    if model.model_type == "LMC":
        mods = ["covar_module.covar_module_list.%d." % i for i in range(model.n_latents)]
    else:
        mods = ["covar_module."]
    ell = torch.cat([gm.softplus(sd[m + "data_covar_module.raw_lengthscale"]).reshape(1, -1) for m in mods], 0)
    B = torch.stack([sd[m + "task_covar_module.covar_factor"] @ sd[m + "task_covar_module.covar_factor"].T
                     + torch.diag_embed(gm.softplus(sd[m + "task_covar_module.raw_var"])) for m in mods])
    p = B.shape[-1]
    if lik.rank == 0:
        S = torch.diag_embed(gm.softplus(sd["lik.raw_task_noises"]) + 1e-4)
    else:
        S = sd["lik.task_noise_covar_factor"] @ sd["lik.task_noise_covar_factor"].T
    S = S + (gm.softplus(sd["lik.raw_noise"]).reshape(()) + 1e-4) * torch.eye(p, dtype=torch.float64)
    means = [k for k in sd if k.startswith("mean_module")]
    mc = torch.cat([sd[k].reshape(-1) for k in sorted(means, key=lambda s: int(s.split(".")[2]))]) if means else None
    return sd, ell, B, S, mc


@pytest.mark.parametrize("model_type,kernel,rank,d", [("LMC", "RBFKernel", 0, 3), ("LMC", "MaternKernel", 2, 3),
                                                      ("ICM", "RBFKernel", 0, 3),
                                                      # more than 8 / more than 16 input dimensions: the gradient
                                                      # epilogue takes the lengthscale sums 8 dimensions per tile walk
                                                      ("LMC", "MaternKernel", 0, 12), ("LMC", "RBFKernel", 2, 20)])
def test_exact_lmc_mll_and_gradients(plmc, model_type, kernel, rank, d):
    n, p, q = 70, 4, 2
    X, Y = _data(n, d, p, seed=5)
    torch.manual_seed(3)
    lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=rank)
    model = plmc.MultitaskGPModel(X, Y, lik, n_tasks=p, n_latents=q, model_type=model_type, init_lmc_coeffs=True,
                                  mean_type=plmc.ConstantMean, kernel_type=getattr(plmc, kernel))
    model, lik = model.double(), lik.double()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for prm in list(model.parameters()) + list(lik.parameters()):
            prm.add_(0.2 * torch.randn(prm.shape, generator=g, dtype=torch.float64))
    sd, ell, B, S, mc = _oracle_inputs(model, lik)
    okind, nu = ("rbf", 2.5) if kernel == "RBFKernel" else ("matern", 2.5)
    ref = ld.lmc_exact_mll(okind, X, Y, ell, B, S, mean_const=mc, nu=nu)
    ref.backward()
    assert torch.allclose(model.lmc_coefficients().double(),
                          torch.stack([sd[k].detach().reshape(-1) for k in sd if k.endswith("covar_factor") and "noise" not in k])
                          if model_type == "LMC" else sd["covar_module.task_covar_module.covar_factor"].detach().T)

    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    mll = plmc.ExactMarginalLogLikelihood(lik, model)
    out = mll(model(X.to(DEV)), Y.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-9 * abs(float(ref)), (float(out), float(ref))
    named = dict(list(model.named_parameters()) + [("lik." + k, v) for k, v in lik.named_parameters()])
    for name, leaf in sd.items():
        got = named[name].grad
        assert got is not None, name
        assert leaf.grad is not None, "oracle did not use %s" % name
        assert torch.allclose(got.cpu(), leaf.grad, rtol=5e-6, atol=1e-9), (name, got.cpu(), leaf.grad)

    # eval-mode prediction (mean and observed variance) against dense conditioning
    Xs = 2 * torch.rand(9, d, dtype=torch.float64) - 1
    mu, var = ld.lmc_posterior(okind, X, Y, Xs, ell.detach(), B.detach(), S.detach(), mean_const=mc.detach(), nu=nu)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model(Xs.to(DEV)))
    assert torch.allclose(pred.mean.cpu(), mu, rtol=1e-8, atol=1e-10)
    assert torch.allclose(pred.variance.cpu(), var + torch.diagonal(S.detach())[None, :], rtol=1e-7, atol=1e-10)


def test_exact_lmc_fp32_and_dense_evaluate(plmc):
    n, d, p, q = 200, 4, 3, 2
    X, Y = _data(n, d, p, seed=8)
    torch.manual_seed(0)
    lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0)
    model = plmc.MultitaskGPModel(X.float(), Y.float(), lik, n_tasks=p, n_latents=q, model_type="LMC",
                                  mean_type=plmc.ZeroMean, kernel_type=plmc.RBFKernel)
    sd, ell, B, S, mc = _oracle_inputs(model, lik)
    ref = float(ld.lmc_exact_mll("rbf", X, Y, ell.detach(), B.detach(), S.detach()))
    model, lik = model.to(DEV), lik.to(DEV)
    model.train()
    val = float(plmc.ExactMarginalLogLikelihood(lik, model)(model(X.float().to(DEV)), Y.float().to(DEV)))
    assert abs(val - ref) < 1e-4 * abs(ref), (val, ref)
    dense = lik(model(X.float().to(DEV))).covariance_matrix.cpu().double()
    C = ld.lmc_covariance("rbf", X, ell.detach(), B.detach(), S.detach())
    assert torch.allclose(dense, C, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("model_type", ["ICM", "LMC"])
def test_multitask_compute_loo(plmc, model_type):
    """MultitaskGPModel.compute_loo (projected_lmc.py:642-656) against the dense formulas on the oracle's covariance:
    sigma2 = 1 / diag(K^-1), y - mu_loo = K^-1 (y - m) sigma2, reshaped to (n, p)."""
    n, d, p, q = 90, 3, 4, 2
    X, Y = _data(n, d, p, seed=21)
    torch.manual_seed(4)
    lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0)
    model = plmc.MultitaskGPModel(X, Y, lik, n_tasks=p, n_latents=q, model_type=model_type, init_lmc_coeffs=True,
                                  mean_type=plmc.ConstantMean, kernel_type=plmc.MaternKernel)
    model, lik = model.double(), lik.double()
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for prm in list(model.parameters()) + list(lik.parameters()):
            prm.add_(0.2 * torch.randn(prm.shape, generator=g, dtype=torch.float64))
    sd, ell, B, S, mc = _oracle_inputs(model, lik)
    with torch.no_grad():
        C = ld.lmc_covariance("matern", X, ell, B, S, 2.5)
        Kinv = torch.cholesky_inverse(torch.linalg.cholesky(C))
        s2_ref = 1.0 / torch.diagonal(Kinv)
        r_ref = (Kinv @ (Y - mc.reshape(1, p)).reshape(-1)) * s2_ref
    model, lik = model.to(DEV), lik.to(DEV)
    s2, r = model.compute_loo()
    assert s2.shape == (n, p) and r.shape == (n, p)
    assert torch.allclose(s2.cpu().reshape(-1), s2_ref, rtol=1e-8, atol=1e-12)
    assert torch.allclose(r.cpu().reshape(-1), r_ref, rtol=1e-7, atol=1e-10)


def test_icm_compute_var(plmc):
    """MultitaskGPModel.compute_var (projected_lmc.py:591-640, ICM only): predictive variance including the task
    noise, clamped at 1e-6 -- the reference gets it from a Kronecker eigen-decomposition, here from the augmented
    factorisation; both equal dense conditioning.  LMC models raise, as in the reference (:600-601)."""
    n, d, p, q = 80, 2, 3, 2
    X, Y = _data(n, d, p, seed=31)
    torch.manual_seed(6)
    lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0)
    model = plmc.MultitaskGPModel(X, Y, lik, n_tasks=p, n_latents=q, model_type="ICM", init_lmc_coeffs=True,
                                  mean_type=plmc.ConstantMean, kernel_type=plmc.RBFKernel)
    model, lik = model.double(), lik.double()
    sd, ell, B, S, mc = _oracle_inputs(model, lik)
    Xs = 2 * torch.rand(11, d, dtype=torch.float64) - 1
    with torch.no_grad():
        _, var = ld.lmc_posterior("rbf", X, Y, Xs, ell, B, S, mean_const=mc, nu=2.5)
        ref = torch.clamp(var + torch.diagonal(S)[None, :], min=1e-6)
    model, lik = model.to(DEV), lik.to(DEV)
    model.train()
    got = model.compute_var(Xs.to(DEV))
    assert model.training                                   # mode restored
    assert got.shape == (11, p) and torch.allclose(got.cpu(), ref, rtol=1e-7, atol=1e-10)
    lmc = plmc.MultitaskGPModel(X, Y, plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0), n_tasks=p, n_latents=q,
                                model_type="LMC", kernel_type=plmc.RBFKernel).double().to(DEV)
    with pytest.raises(ValueError, match="only available for ICM"):
        lmc.compute_var(Xs.to(DEV))
