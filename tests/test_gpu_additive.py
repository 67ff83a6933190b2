"""GPU parity of additive sub-kernel decompositions (`decomp`, projected_lmc.py:131-167; SURVEY.md
8f row 4) for a single-output ExactGPModel: MLL, gradients, prediction."""
import pytest
import torch

from oracle import gp_math as gm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_additive_decomposition_exact_gp():
    import projectedlmc as plmc
    g = torch.Generator().manual_seed(0)
    n, d = 180, 3
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    y = torch.randn(n, generator=g, dtype=torch.float64)
    decomp = [[0, 1], [1, 2]]
    lik = plmc.GaussianLikelihood()
    model = plmc.ExactGPModel(X, y, lik, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel, decomp=decomp)
    model, lik = model.double(), lik.double()
    gg = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for prm in model.parameters():
            prm.add_(0.3 * torch.randn(prm.shape, generator=gg, dtype=torch.float64))
    leaves = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    K = torch.zeros(n, n, dtype=torch.float64)
    for gi, idx in enumerate(decomp):
        ell = gm.softplus(leaves["covar_module.kernels.%d.base_kernel.raw_lengthscale" % gi]).reshape(1, -1)
        os_ = gm.softplus(leaves["covar_module.kernels.%d.raw_outputscale" % gi]).reshape(1)
        K = K + gm.kernel_matrix("matern", X[:, idx], X[:, idx], ell, os_, 2.5)[0]
    noise = gm.softplus(leaves["likelihood.noise_covar.raw_noise"]).reshape(()) + 1e-4
    ref = gm.mvn_log_prob(K + noise * torch.eye(n, dtype=torch.float64), y) / n
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    out = plmc.ExactMarginalLogLikelihood(lik, model)(model(X.to(DEV)), y.to(DEV)).sum()
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-9 * abs(float(ref)), (float(out), float(ref))
    for name, prm in model.named_parameters():
        assert torch.allclose(prm.grad.cpu(), leaves[name].grad, rtol=1e-5, atol=1e-9), name
    assert len(model.lscales()) == 2 and model.outputscale().shape == (1, 2)
    # prediction against dense conditioning
    Xs = 2 * torch.rand(20, d, dtype=torch.float64) - 1
    with torch.no_grad():
        Ks = torch.zeros(20, n, dtype=torch.float64)
        for gi, idx in enumerate(decomp):
            ell = gm.softplus(leaves["covar_module.kernels.%d.base_kernel.raw_lengthscale" % gi]).reshape(1, -1)
            os_ = gm.softplus(leaves["covar_module.kernels.%d.raw_outputscale" % gi]).reshape(1)
            Ks = Ks + gm.kernel_matrix("matern", Xs[:, idx], X[:, idx], ell, os_, 2.5)[0]
        mu = Ks @ torch.linalg.solve(K.detach() + noise.detach() * torch.eye(n, dtype=torch.float64), y)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = model(Xs.to(DEV))
    assert torch.allclose(pred.mean.cpu(), mu, rtol=1e-7, atol=1e-9)


def test_additive_decomposition_batched_latents():
    """`decomp` with batch_shape = [n_funcs] (projected_lmc.py:151-167): a batch of q independent GPs whose kernels
    are sums of scaled sub-kernels on subsets of the inputs -- what the factory builds for the latent processes of a
    ProjectedGPModel / a batched ExactGPModel.  MLL, every gradient and the posterior against dense formulas."""
    import projectedlmc as plmc
    g = torch.Generator().manual_seed(3)
    n, d, q = 150, 3, 3
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(q, n, generator=g, dtype=torch.float64)
    decomp = [[0, 1], [2]]
    lik = plmc.GaussianLikelihood(batch_shape=torch.Size([q]))
    model = plmc.ExactGPModel(X, Y, lik, n_tasks=q, mean_type=plmc.ZeroMean, kernel_type=plmc.RBFKernel, decomp=decomp)
    model, lik = model.double(), lik.double()
    gg = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for prm in model.parameters():
            prm.add_(0.3 * torch.randn(prm.shape, generator=gg, dtype=torch.float64))
    leaves = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    noise = gm.softplus(leaves["likelihood.noise_covar.raw_noise"]).reshape(q) + 1e-4

    def K_of(Xa, Xb):
        K = torch.zeros(q, Xa.shape[0], Xb.shape[0], dtype=torch.float64)
        for gi, idx in enumerate(decomp):
            ell = gm.softplus(leaves["covar_module.kernels.%d.base_kernel.raw_lengthscale" % gi]).reshape(q, -1)
            os_ = gm.softplus(leaves["covar_module.kernels.%d.raw_outputscale" % gi]).reshape(q)
            K = K + gm.kernel_matrix("rbf", Xa[:, idx], Xb[:, idx], ell, os_, 2.5)
        return K
    K = K_of(X, X) + noise.reshape(q, 1, 1) * torch.eye(n, dtype=torch.float64)
    ref = gm.mvn_log_prob(K, Y).sum() / n
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    out = plmc.ExactMarginalLogLikelihood(lik, model)(model(X.to(DEV)), Y.to(DEV)).sum()
    out.backward()
    assert abs(float(out.detach()) - float(ref)) < 1e-9 * abs(float(ref)), (float(out), float(ref))
    for name, prm in model.named_parameters():
        assert torch.allclose(prm.grad.cpu(), leaves[name].grad, rtol=1e-5, atol=1e-9), name
    Xs = 2 * torch.rand(15, d, dtype=torch.float64) - 1
    with torch.no_grad():
        mu = (K_of(Xs, X) @ torch.linalg.solve(K, Y.unsqueeze(-1))).squeeze(-1)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = model(Xs.to(DEV))
    assert torch.allclose(pred.mean.cpu().reshape(q, -1), mu, rtol=1e-7, atol=1e-9)
