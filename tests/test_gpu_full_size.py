"""BASELINE configs at their FULL sizes under pytest: C2 (exact LMC, n = 2048, 8 tasks, 4 latents,
fp64: one 16384 x 16384 matrix), an intermediate dense-LMC size that runs the multi-group look-ahead schedule against
the oracle with every gradient, and C4 (SVGP-LMC, 16 tasks, 2000 inducing points, fp32).  Where the dense fp64 oracle
is affordable on the box's host cores it is the reference (value at C2 and C4); gradients at full size are tied down
by exact identities of the Gaussian log-density (homogeneity / Euler relations, scaling, permutation)."""
import math
import warnings

import pytest
import torch

from oracle import gp_math as gm
from oracle import lmc_dense as ld
from test_gpu_multitask import _data, _oracle_inputs
from test_gpu_variational import _build, _oracle_elbo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def plmc():
    import projectedlmc
    assert torch.cuda.is_available()
    return projectedlmc


def test_dense_lmc_multi_group_schedule_against_oracle(plmc):
    """n = 520, p = 4 (N = 2080: 17 block rows, three groups -> the look-ahead schedule, k_lmc_kinv_grad on a
    multi-group matrix): MLL and the gradient of every parameter against the fp64 oracle."""
    n, d, p, q = 520, 3, 4, 2
    X, Y = _data(n, d, p, seed=11)
    torch.manual_seed(5)
    lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0)
    model = plmc.MultitaskGPModel(X, Y, lik, n_tasks=p, n_latents=q, model_type="LMC", init_lmc_coeffs=True,
                                  mean_type=plmc.ConstantMean, kernel_type=plmc.MaternKernel)
    model, lik = model.double(), lik.double()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for prm in list(model.parameters()) + list(lik.parameters()):
            prm.add_(0.2 * torch.randn(prm.shape, generator=g, dtype=torch.float64))
    sd, ell, B, S, mc = _oracle_inputs(model, lik)
    ref = ld.lmc_exact_mll("matern", X, Y, ell, B, S, mean_const=mc, nu=2.5)
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    out = plmc.ExactMarginalLogLikelihood(lik, model)(model(X.to(DEV)), Y.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-9 * abs(float(ref)), (float(out), float(ref))
    named = dict(list(model.named_parameters()) + [("lik." + k, v) for k, v in lik.named_parameters()])
    for name, leaf in sd.items():
        assert named[name].grad is not None and leaf.grad is not None, name
        assert torch.allclose(named[name].grad.cpu(), leaf.grad, rtol=5e-6, atol=1e-9), name


def test_config2_full_size_fp64(plmc):
    """C2 itself: n = 2048, p = 8, q = 4, RBF, fp64, N = n p = 16384 (128 block rows).
    (1) value against the dense fp64 oracle (one 16384^2 Cholesky on the host cores);
    (2) Euler identity of the gradient kernel: K is linear in (B_1..B_q, Sigma), so
        sum_i <B_i, dlogp/dB_i> + <Sigma, dlogp/dSigma> = (quad - N) / 2,  quad = -y . dlogp/dy;
    (3) scaling: logp(c y; c^2 B, c^2 Sigma) = logp(y; B, Sigma) - N log c;
    (4) linearity in y: logp(y / 2) - logp(y) = 3/8 quad."""
    from projectedlmc import _lmc_engine
    n, d, p, q = 2048, 8, 8, 4
    N = n * p
    g = torch.Generator().manual_seed(2)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(n, p, generator=g, dtype=torch.float64)
    ell = 0.6 + 0.8 * torch.rand(q, d, generator=g, dtype=torch.float64)
    F = torch.randn(q, p, 1, generator=g, dtype=torch.float64)
    B = F @ F.transpose(-1, -2) + torch.diag_embed(0.05 + 0.2 * torch.rand(q, p, generator=g, dtype=torch.float64))
    Sigma = torch.diag_embed(0.05 + 0.3 * torch.rand(p, generator=g, dtype=torch.float64))
    f = lambda t: t.to(DEV)
    Bd, Sd, yd = f(B).requires_grad_(), f(Sigma).requires_grad_(), f(Y.reshape(-1)).requires_grad_()
    lp = _lmc_engine.lmc_exact_log_prob("rbf", f(X), f(ell), None, Bd, Sd, yd)
    lp.backward()
    quad = -float((yd.grad * yd.detach()).sum())
    assert quad > 0
    # (2)
    euler = float((Bd.grad * Bd.detach()).sum() + (Sd.grad * Sd.detach()).sum())
    assert abs(euler - 0.5 * (quad - N)) < 1e-8 * N, (euler, 0.5 * (quad - N))
    # (3), (4)
    c = 1.7
    lp_c = _lmc_engine.lmc_exact_log_prob("rbf", f(X), f(ell), None, f(B) * c * c, f(Sigma) * c * c, f(Y.reshape(-1)) * c)
    assert abs(float(lp_c) - (float(lp) - N * math.log(c))) < 1e-9 * abs(float(lp))
    lp_h = _lmc_engine.lmc_exact_log_prob("rbf", f(X), f(ell), None, f(B), f(Sigma), 0.5 * f(Y.reshape(-1)))
    assert abs(float(lp_h) - float(lp) - 0.375 * quad) < 1e-9 * abs(float(lp))
    # (1) dense oracle value (host: ~1.5e12 flop)
    ref = float(ld.lmc_exact_mll("rbf", X, Y, ell, B, Sigma)) * N
    assert abs(float(lp) - ref) < 1e-9 * abs(ref), (float(lp), ref)


def test_config4_full_size_fp32(plmc):
    """C4 itself: 16 tasks, 8 latents, 2000 inducing points (n = 3000 training points, train_ind_ratio 1.5), Cholesky
    variational distribution, fp32.  ELBO against the fp64 oracle evaluated at the same (fp32-rounded) parameters
    (1e-4 relative, the BASELINE tolerance), gradients of the kernel / variational / mixing parameters against the
    oracle's autograd (2e-3 of the largest entry), and the prior identity: with q(u) at its initial value (the
    whitened prior) the KL term vanishes and the predictive marginals are the prior's."""
    n, d, p, q = 3000, 8, 16, 8
    X, Y, model, lik = _build(plmc, n, d, p, q, "RBFKernel", False, torch.float32, seed=4)
    assert model.variational_strategy.base_variational_strategy.inducing_points.shape[-2] == 2000
    ref, sd = _oracle_elbo(model, lik, X, Y, "rbf", 2.5, 1e-4)
    ref.backward()
    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    out = plmc.VariationalELBO(lik, model, num_data=n)(model(X.to(DEV)), Y.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-4 * abs(float(ref)), (float(out), float(ref))
    named = dict(list(model.named_parameters()) + [("lik." + k, v) for k, v in lik.named_parameters()])
    for name, leaf in sd.items():
        if leaf.grad is None:
            continue
        got, ref_g = named[name].grad.cpu().double(), leaf.grad
        if name.endswith("chol_variational_covar"):
            got, ref_g = got.tril(), ref_g.tril()
        scale = float(ref_g.abs().max()) + 1e-12
        assert float((got - ref_g).abs().max()) < 2e-3 * scale, (name, float((got - ref_g).abs().max()), scale)


def test_sharded_prediction_sums_to_the_unsharded_posterior(plmc):
    """Sharded eval path (projected_lmc.py:1144,1152 is where the cross-latent sum happens): the partial task means and
    variances of the latent shards (one process here, so the all-reduce is the identity) add up to the unsharded
    posterior; each shard carries the eps of :1153 once."""
    n, d, p, q, ns = 300, 3, 6, 4, 40
    g = torch.Generator().manual_seed(9)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(n, p, generator=g, dtype=torch.float64)
    Xs = 2 * torch.rand(ns, d, generator=g, dtype=torch.float64) - 1
    torch.manual_seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = plmc.ProjectedGPModel(X, Y, p, q, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel, init_lmc_coeffs=True,
                                  BDN=False).double().to(DEV)
    m.eval()
    with torch.no_grad():
        full = m(Xs.to(DEV))
        mean_ref, var_ref = full.mean.clone(), full.variance.clone()
        world = 2
        mean_sum, var_sum = torch.zeros_like(mean_ref), torch.zeros_like(var_ref)
        for r in range(world):
            m.set_latent_shard((r, world))
            part = m(Xs.to(DEV))
            mean_sum += part.mean
            var_sum += part.variance
        m.set_latent_shard(None)
    assert torch.allclose(mean_sum, mean_ref, rtol=1e-10, atol=1e-12)
    assert torch.allclose(var_sum - (world - 1) * m.eps, var_ref, rtol=1e-9, atol=1e-12)
