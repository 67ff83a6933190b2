"""HIP path against the committed golden vectors (tests/golden/projected_v1.json): training loss, every
parameter gradient, and eval-mode predictions of the projected model; MLL, gradients and posterior of the
single-output exact GP.  fp64 engine, tolerances 1e-9 (values) / 2e-6 (gradients)."""
import pytest
import torch

from _golden import cases, T, build_projected

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("c", cases("projected"), ids=lambda c: c["name"])
def test_projected_model_matches_golden(c):
    import projectedlmc as plmc
    m, X, Y = build_projected(plmc, c)
    m = m.to(DEV)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    m.train(); m.likelihood.train()
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    loss = -mll(m(Xd), Yd)
    loss.backward()
    assert abs(float(loss.detach()) - c["loss"]) <= 1e-9 * abs(c["loss"])
    named = dict(m.named_parameters())
    for pname, g in c["grads"].items():
        got = named[pname].grad.cpu().reshape(-1)
        ref = T(g).reshape(-1)
        if pname.endswith("B_tilde_inv_chol.original"):              # only the lower triangle is a free parameter
            sh = named[pname].shape
            got, ref = got.reshape(sh).tril().reshape(-1), ref.reshape(sh).tril().reshape(-1)
        assert torch.allclose(got, ref, rtol=2e-6, atol=1e-9 * (1 + float(ref.abs().max()))), pname
    m.eval(); m.likelihood.eval()
    with torch.no_grad():
        full_likelihood = m.full_likelihood()
        dist = m(T(c["Xs"]).to(DEV))
        obs = full_likelihood(dist)
    assert torch.allclose(dist.mean.cpu(), T(c["pred_mean"]), rtol=1e-8, atol=1e-10)
    assert torch.allclose(dist.variance.cpu(), T(c["pred_var"]), rtol=1e-7, atol=1e-10)
    assert torch.allclose(obs.variance.cpu(), T(c["pred_var_observed"]), rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("c", cases("exact"), ids=lambda c: c["name"])
def test_exact_gp_matches_golden(c):
    import projectedlmc as plmc
    X, y = T(c["X"]), T(c["y"])
    lik = plmc.GaussianLikelihood().double()
    model = plmc.ExactGPModel(X, y, lik, mean_type=plmc.ZeroMean, kernel_type=getattr(plmc, c["kernel"]),
                              outputscales=False).double()
    with torch.no_grad():
        model.covar_module.raw_lengthscale.copy_(T(c["raw_lengthscale"]).reshape(model.covar_module.raw_lengthscale.shape))
        lik.noise_covar.raw_noise.copy_(T(c["raw_noise"]).reshape(lik.noise_covar.raw_noise.shape))
    model = model.to(DEV); lik = lik.to(DEV)
    model.train(); lik.train()
    mll = plmc.ExactMarginalLogLikelihood(lik, model)
    out = mll(model(X.to(DEV)), y.to(DEV))
    out.backward()
    assert abs(float(out.detach()) - c["mll"]) <= 1e-9 * abs(c["mll"])
    assert torch.allclose(model.covar_module.raw_lengthscale.grad.cpu().reshape(-1), T(c["grad_raw_lengthscale"]).reshape(-1), rtol=2e-6, atol=1e-10)
    assert torch.allclose(lik.noise_covar.raw_noise.grad.cpu().reshape(-1), T(c["grad_raw_noise"]).reshape(-1), rtol=2e-6, atol=1e-10)
    model.eval(); lik.eval()
    with torch.no_grad():
        post = model(T(c["Xs"]).to(DEV))
    assert torch.allclose(post.mean.cpu().reshape(-1), T(c["pred_mean"]), rtol=1e-8, atol=1e-10)
    assert torch.allclose(post.variance.cpu().reshape(-1), T(c["pred_var"]), rtol=1e-7, atol=1e-10)
