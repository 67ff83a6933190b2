"""The reference's SplineKernel (projected_lmc.py:26-36) on the HIP path -- kernel kind "spline" of the assembly, cross and
gradient kernels: exact single-output GP (MLL, every gradient, posterior mean / variance with the non-constant prior
diagonal) and the latent processes of a projected model, against dense formulas on the oracle's restatement of the
reference's own `forward`."""
import warnings

import pytest
import torch

from oracle import gp_math as gm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,d", [(150, 3), (300, 11), (200, 19)])
def test_spline_kernel_exact_gp(n, d):
    import projectedlmc as plmc
    g = torch.Generator().manual_seed(0)
    X = torch.rand(n, d, generator=g, dtype=torch.float64)                    # the spline kernel lives on x >= 0
    y = torch.randn(n, generator=g, dtype=torch.float64)
    lik = plmc.GaussianLikelihood()
    model = plmc.ExactGPModel(X, y, lik, mean_type=plmc.ConstantMean, kernel_type=plmc.SplineKernel, outputscales=True)
    model, lik = model.double(), lik.double()
    gg = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for prm in model.parameters():
            prm.add_(0.3 * torch.randn(prm.shape, generator=gg, dtype=torch.float64))
    leaves = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    os_ = gm.softplus(leaves["covar_module.raw_outputscale"]).reshape(1)
    noise = gm.softplus(leaves["likelihood.noise_covar.raw_noise"]).reshape(()) + 1e-4
    mean = leaves["mean_module.raw_constant"].reshape(())
    one = torch.ones(1, d, dtype=torch.float64)
    K = gm.kernel_matrix("spline", X, X, one, os_)[0] + noise * torch.eye(n, dtype=torch.float64)
    ref = gm.mvn_log_prob(K, y - mean) / n
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    out = plmc.ExactMarginalLogLikelihood(lik, model)(model(X.to(DEV)), y.to(DEV)).sum()
    out.backward()
    assert abs(float(out.detach()) - float(ref.detach())) < 1e-9 * abs(float(ref)), (float(out.detach()), float(ref.detach()))
    for name, prm in model.named_parameters():
        assert torch.allclose(prm.grad.cpu(), leaves[name].grad, rtol=1e-5, atol=1e-9), name
    Xs = torch.rand(25, d, dtype=torch.float64)
    with torch.no_grad():
        Ks = gm.kernel_matrix("spline", Xs, X, one, os_)[0]
        sol = torch.linalg.solve(K, torch.cat([(y - mean)[:, None], Ks.T], 1))
        mu = mean + Ks @ sol[:, 0]
        var = os_ * (1 + Xs ** 2 + Xs ** 3 / 3).prod(-1) - (Ks * sol[:, 1:].T).sum(-1)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = model(Xs.to(DEV))
    assert torch.allclose(pred.mean.cpu().reshape(-1), mu, rtol=1e-7, atol=1e-9)
    assert torch.allclose(pred.variance.cpu().reshape(-1), var, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("d", [4, 13, 25])
def test_spline_kernel_projected_latents_fp32(d):
    """q latent GPs with the spline kernel inside the projected model (fp32): the latent log-likelihood term the HIP
    engine computes equals the dense formula on the projected data."""
    import projectedlmc as plmc
    from projectedlmc import _engine
    g = torch.Generator().manual_seed(2)
    n, q = 700, 3
    X = torch.rand(n, d, generator=g, dtype=torch.float64) * (4.0 / d) ** 0.5     # keeps the product of d factors O(1)
    yt = torch.randn(q, n, generator=g, dtype=torch.float64)
    noise = torch.tensor([0.2, 0.4, 0.6], dtype=torch.float64)
    osc = torch.tensor([0.7, 1.0, 1.3], dtype=torch.float64)
    one = torch.ones(q, d, dtype=torch.float64)
    K = gm.kernel_matrix("spline", X, X, one, osc) + noise.reshape(q, 1, 1) * torch.eye(n, dtype=torch.float64)
    ref = gm.mvn_log_prob(K, yt)
    f = lambda t: t.to(DEV, torch.float32)
    nz = f(noise).requires_grad_()
    osd = f(osc).requires_grad_()
    lp = _engine.exact_latent_log_prob("spline", f(X), f(one), osd, nz, f(yt))
    lp.sum().backward()
    assert float(((lp.detach().cpu().double() - ref) / ref).abs().max()) < 1e-4
    Kinv = torch.cholesky_inverse(torch.linalg.cholesky(K))
    alpha = (Kinv @ yt.unsqueeze(-1)).squeeze(-1)
    g_noise = 0.5 * ((alpha * alpha).sum(-1) - torch.diagonal(Kinv, dim1=-2, dim2=-1).sum(-1))
    assert float(((nz.grad.cpu().double() - g_noise).abs() / g_noise.abs()).max()) < 2e-3
    Kos = gm.kernel_matrix("spline", X, X, one, None)
    g_os = 0.5 * (((alpha.unsqueeze(-1) * alpha.unsqueeze(-2)) - Kinv) * Kos).sum((-2, -1))
    assert float(((osd.grad.cpu().double() - g_os).abs() / g_os.abs()).max()) < 2e-3
