"""N > 1 path on CPU: world_size-2 `gloo` processes exercising latent sharding
(SURVEY.md 8e): partition of the latents, rank-local loss shares, the single fused all-reduce of
loss + gradients, and identical optimiser steps on every rank.

The n x n log-likelihood itself needs the GPU, so inside the worker processes the engine call is
replaced by the oracle's torch implementation (test-only stand-in); everything else -- model,
ProjectedLMCmll, parallel.sync_loss_and_grads -- is the product code."""
import os
import sys
import warnings

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, ret):
    for p in (ROOT, os.path.join(ROOT, "projected-lmc_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_default_dtype(torch.float64)
    import projectedlmc as plmc
    from projectedlmc import _engine, parallel
    from oracle import gp_math as gm
    kinds = {"rbf": ("rbf", 2.5), "matern52": ("matern", 2.5)}

    def fake_log_prob(kind, X, ell, oscale, noise, y, hyper=None):     # oracle stand-in for the HIP call
        k, nu = kinds[kind]
        return gm.exact_latent_log_prob(k, X, ell, noise, y, oscale, nu)
    _engine.exact_latent_log_prob = fake_log_prob
    from projectedlmc import distributions
    distributions.MultivariateNormal.log_prob = lambda self, value: fake_log_prob(
        self._covar.kind, self._covar.x1, self._covar.ell, self._covar.oscale, self._covar.noise.reshape(-1),
        (value - self.loc).reshape(self._covar.ell.shape[0], -1))

    g = torch.Generator().manual_seed(0)
    n, d, p = 48, 2, 5
    X = 2 * torch.rand(n, d, generator=g) - 1
    Y = torch.randn(n, p, generator=g)

    def build(shard):
        torch.manual_seed(1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = plmc.ProjectedGPModel(X, Y, p, q, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                      init_lmc_coeffs=True, BDN=False, latent_shard=shard)
        m.train()
        return m, plmc.ProjectedLMCmll(m.likelihood, m)

    # reference: un-sharded model (computed identically on every rank)
    m0, mll0 = build(None)
    loss0 = -mll0(m0(X), Y)
    loss0.backward()
    # sharded
    m1, mll1 = build(parallel.shard_of())
    assert m1.latent_ids == list(range(rank, q, world))
    share = -mll1(m1(X), Y)
    share.backward()
    total = parallel.sync_loss_and_grads(share, list(m1.parameters()))
    ok = abs(float(total) - float(loss0)) < 1e-10 * abs(float(loss0))
    for (na, a), (nb, b) in zip(m0.named_parameters(), m1.named_parameters()):
        ok = ok and torch.allclose(a.grad, b.grad, rtol=1e-9, atol=1e-12)
    # identical steps keep replicas in sync
    opt = torch.optim.AdamW(m1.parameters(), lr=1e-2)
    opt.step()
    flat = torch.cat([p_.detach().reshape(-1) for p_ in m1.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)

    # sharded PREDICTION (projected_lmc.py:1144,1152): every rank computes the posterior of its own latents, one
    # all-reduce of the (2, n*, p) partial mean / variance sums forms the task-space posterior on every rank
    def fake_posterior(kind, X_, ell, oscale, noise, y, Xs_, full_cov=False, cache=None, key=None):
        k, nu = kinds[kind]
        mu, cov = gm.exact_gp_posterior(k, X_, ell, noise, y, Xs_, oscale, nu)
        return mu, (cov if full_cov else torch.diagonal(cov, dim1=-2, dim2=-1))
    _engine.exact_posterior = fake_posterior
    # ... and of the mixing kernel (plmc_mix_posterior): the host logic under test is the shard / all-reduce around it
    _engine.mix_posterior = lambda ml, vl, Ht, eps=0.0: (ml.T @ Ht, vl.T @ (Ht * Ht) + eps)
    Xs = 2 * torch.rand(9, d, generator=torch.Generator().manual_seed(5)) - 1
    m0.eval(); m1.eval()
    with torch.no_grad():
        for pa, pb in zip(m0.parameters(), m1.parameters()):          # same parameters on both models
            pa.copy_(pb)
        ref = m0(Xs)
        got = m1(Xs)
    ok = ok and torch.allclose(got.mean, ref.mean, rtol=1e-10, atol=1e-12)
    ok = ok and torch.allclose(got.variance, ref.variance, rtol=1e-9, atol=1e-12)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("q", [2, 3])
def test_latent_sharding_world2_gloo(q):
    world = 2
    port = 29500 + (os.getpid() % 2000) + q
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, q, ret), nprocs=world, join=True)
    assert all(ret.get(r, False) for r in range(world)), dict(ret)
