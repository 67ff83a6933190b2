"""plmc_posterior_moments / plmc_mix_posterior (include/plmc.h; projected_lmc.py:1133-1155): the reductions behind the
augmented sweep of the eval-mode posterior, against dense fp64 formulas on the oracle's covariance -- latent moments from
[Khat | y | K*^T], task-space mixing, and the whole ProjectedGPModel prediction through them."""
import pytest
import torch

from oracle import gp_math as gm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,ns,q,dtype", [(300, 37, 3, torch.float64), (1100, 130, 2, torch.float64), (700, 65, 4, torch.float32)])
def test_posterior_moments_match_dense_conditioning(n, ns, q, dtype):
    from projectedlmc import _engine
    g = torch.Generator().manual_seed(n)
    d = 3
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Xs = 2 * torch.rand(ns, d, generator=g, dtype=torch.float64) - 1
    y = torch.randn(q, n, generator=g, dtype=torch.float64)
    ell = 0.5 + 0.5 * torch.rand(q, d, generator=g, dtype=torch.float64)
    noise = 0.1 + 0.3 * torch.rand(q, generator=g, dtype=torch.float64)
    osc = 0.5 + torch.rand(q, generator=g, dtype=torch.float64)
    mu, cov = gm.exact_gp_posterior("matern", X, ell, noise, y, Xs, osc, 2.5)
    f = lambda t: t.to(DEV, dtype)
    mean, var = _engine.exact_posterior("matern52", f(X), f(ell), f(osc), f(noise), f(y), f(Xs))
    rt, at = (1e-9, 1e-10) if dtype == torch.float64 else (2e-3, 2e-4)
    assert torch.allclose(mean.cpu().double(), mu, rtol=rt, atol=at)
    assert torch.allclose(var.cpu().double(), torch.diagonal(cov, dim1=-2, dim2=-1), rtol=rt, atol=at)
    mean_f, cov_f = _engine.exact_posterior("matern52", f(X), f(ell), f(osc), f(noise), f(y), f(Xs), full_cov=True)
    assert torch.equal(mean_f, mean)
    assert torch.allclose(cov_f.cpu().double(), cov, rtol=rt, atol=at)


@pytest.mark.parametrize("q,ns,p,dtype", [(3, 50, 7, torch.float64), (8, 1000, 16, torch.float32), (1, 5, 1, torch.float64)])
def test_mix_posterior_matches_matmul(q, ns, p, dtype):
    from projectedlmc import _engine
    g = torch.Generator().manual_seed(q * 100 + p)
    ml = torch.randn(q, ns, generator=g, dtype=torch.float64)
    vl = torch.rand(q, ns, generator=g, dtype=torch.float64)
    Ht = torch.randn(q, p, generator=g, dtype=torch.float64)
    eps = 1e-3
    f = lambda t: t.to(DEV, dtype)
    mean, var = _engine.mix_posterior(f(ml), f(vl), f(Ht), eps)
    assert mean.shape == (ns, p) and var.shape == (ns, p)
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    assert torch.allclose(mean.cpu().double(), f(ml).cpu().double().T @ f(Ht).cpu().double(), rtol=tol, atol=tol)
    Hd = f(Ht).cpu().double()
    assert torch.allclose(var.cpu().double(), f(vl).cpu().double().T @ (Hd * Hd) + eps, rtol=tol, atol=tol)
    # a shard's partial sums add up to the whole (sharded prediction adds eps after the all-reduce)
    if q > 1:
        m0, v0 = _engine.mix_posterior(f(ml)[:1], f(vl)[:1], f(Ht)[:1], 0.0)
        m1, v1 = _engine.mix_posterior(f(ml)[1:], f(vl)[1:], f(Ht)[1:], 0.0)
        assert torch.allclose(m0 + m1, mean, rtol=10 * tol, atol=10 * tol)
        assert torch.allclose(v0 + v1 + eps, var, rtol=10 * tol, atol=10 * tol)


def test_mix_posterior_refuses_host_tensors():
    from projectedlmc import _engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _engine.mix_posterior(torch.zeros(2, 3), torch.zeros(2, 3), torch.zeros(2, 4))
