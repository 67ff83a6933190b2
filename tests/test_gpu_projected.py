"""GPU parity of the model-level API (ProjectedGPModel + ProjectedLMCmll, ExactGPModel + ExactMLL)
against the CPU oracle: loss, gradients w.r.t. every parameter, and eval-mode predictions."""
import math
import warnings

import pytest
import torch

from oracle import projected as pj
from oracle import lmc_dense as ld
from oracle import gp_math as gm
from _bridge import oracle_params, param_map, perturb_

pytestmark = pytest.mark.gpu

VARIANTS = {
    "PLMC": dict(BDN=False, diagonal_B=False, scalar_B=False),            # experiments.py:197-200
    "PLMC_diagB": dict(BDN=False, diagonal_B=True, scalar_B=False),
    "BDN_fullB": dict(BDN=True, diagonal_B=False, scalar_B=False),
    "oilmm": dict(BDN=True, diagonal_B=True, scalar_B=True, diagonal_R=True),   # experiments.py:203-207
    "PLMC_fast": dict(BDN=True, diagonal_B=True, scalar_B=True),          # experiments.py:211-215
}
DEV = "cuda:0"


def _data(n, d, p, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(n, p, generator=g, dtype=torch.float64)
    return X, Y


def _model(plmc, X, Y, q, kernel, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return plmc.ProjectedGPModel(X, Y, Y.shape[1], q, mean_type=plmc.ZeroMean, kernel_type=kernel, **kw)


@pytest.fixture(scope="module")
def plmc():
    import projectedlmc
    assert torch.cuda.is_available()
    return projectedlmc


@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("kernel,oscale", [("MaternKernel", False), ("RBFKernel", True)])
def test_training_loss_and_all_gradients(plmc, name, kernel, oscale):
    n, d, p, q = 333, 3, 6, 2
    X, Y = _data(n, d, p, seed=11)
    torch.manual_seed(5)
    m = _model(plmc, X, Y, q, getattr(plmc, kernel), init_lmc_coeffs=True, outputscales=oscale, **VARIANTS[name])
    m = perturb_(m.double())
    P = oracle_params(m)
    for k in pj.tensor_keys(P):
        P[k].requires_grad_(True)
    ref = -pj.projected_mll(P, X, Y)
    ref.backward()

    m = m.to(DEV)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    m.train()
    m.likelihood.train()
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    loss = -mll(m(Xd), Yd)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-9 * abs(float(ref)), (float(loss), float(ref))
    pm = param_map(m)
    checked = 0
    for pname, prm in m.named_parameters():
        g_ref = P[pm[pname]].grad
        assert prm.grad is not None, pname
        assert torch.allclose(prm.grad.cpu(), g_ref, rtol=2e-6, atol=1e-8), (pname, prm.grad.cpu(), g_ref)
        checked += 1
    assert checked >= 4
    # the stored projection terms are those of the reference's proj_term_list (:1206)
    terms, _ = pj.projection_terms(P, Y)
    for a, b in zip(mll.proj_term_list, terms):
        assert abs(float(a) - float(b)) < 1e-9 * max(1.0, abs(float(b)))


NONBULK = {
    # realdata_experiments.py:107-111, the OILMM run: bulk=False, orthogonally parametrised Q (p x q), diagonal positive R
    "oilmm": dict(BDN=True, diagonal_B=True, scalar_B=True, diagonal_R=True),
    "PLMC": dict(BDN=False, diagonal_B=False, scalar_B=False),                       # Q_plus p x p, upper-triangular R
    "BDN_diagB_cayley": dict(BDN=True, diagonal_B=True, scalar_B=False, ortho_param="cayley"),
}


@pytest.mark.parametrize("name", list(NONBULK))
@pytest.mark.parametrize("kernel", ["MaternKernel", "RBFKernel"])
def test_separately_parametrised_mixing_matrix_loss_and_all_gradients(plmc, name, kernel):
    """`bulk=False` (projected_lmc.py:851-853, :873, :884, :963-970; loss branch :1237): loss, every gradient (incl. the raw
    orthogonal and triangular parameters), the stored projection terms and the eval-mode task posterior against the oracle."""
    n, d, p, q = 333, 3, 6, 2
    X, Y = _data(n, d, p, seed=12)
    torch.manual_seed(6)
    m = _model(plmc, X, Y, q, getattr(plmc, kernel), init_lmc_coeffs=True, bulk=False, **NONBULK[name])
    assert not m.lmc_coefficients.bulk
    m = perturb_(m.double())
    P = oracle_params(m)
    assert not P["bulk"] and P["diagonal_R"] == bool(NONBULK[name].get("diagonal_R", False))
    for k in pj.tensor_keys(P):
        P[k].requires_grad_(True)
    ref = -pj.projected_mll(P, X, Y)
    ref.backward()

    m = m.to(DEV)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    m.train()
    m.likelihood.train()
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    loss = -mll(m(Xd), Yd)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-9 * abs(float(ref)), (float(loss), float(ref))
    pm = param_map(m)
    names = [pname for pname, _ in m.named_parameters()]
    assert "lmc_coefficients.parametrizations.Q_plus.original" in names and "lmc_coefficients.parametrizations.R.original" in names
    for pname, prm in m.named_parameters():
        g_ref = P[pm[pname]].grad
        assert prm.grad is not None, pname
        assert torch.allclose(prm.grad.cpu(), g_ref, rtol=2e-6, atol=1e-8), (pname, prm.grad.cpu(), g_ref)
    terms, _ = pj.projection_terms(P, Y)
    for a, b in zip(mll.proj_term_list, terms):
        assert abs(float(a) - float(b)) < 1e-9 * max(1.0, abs(float(b)))
    # eval mode: the task posterior mixes with H = Q R of the parametrised pair (:884)
    Pd = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in P.items()}
    Xs = 2 * torch.rand(48, d, dtype=torch.float64) - 1
    mean_ref, cov_ref = pj.task_posterior(Pd, X, Y, Xs)
    _, var_obs_ref = pj.observed_posterior(Pd, X, Y, Xs)
    m.eval()
    m.likelihood.eval()
    with torch.no_grad():
        dist = m(Xs.to(DEV))
        obs = m.full_likelihood()(dist)
    assert torch.allclose(dist.mean.cpu(), mean_ref, rtol=1e-8, atol=1e-10)
    assert torch.allclose(dist.variance.cpu(), torch.diagonal(cov_ref).reshape(48, p), rtol=1e-7, atol=1e-10)
    assert torch.allclose(obs.variance.cpu(), var_obs_ref, rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("name", ["PLMC", "PLMC_fast"])
def test_eval_mode_task_posterior(plmc, name):
    n, d, p, q, ns = 200, 2, 5, 2, 64
    X, Y = _data(n, d, p, seed=2)
    Xs = 2 * torch.rand(ns, d, dtype=torch.float64) - 1
    torch.manual_seed(1)
    m = perturb_(_model(plmc, X, Y, q, plmc.MaternKernel, init_lmc_coeffs=True, **VARIANTS[name]).double())
    P = oracle_params(m)
    mean_ref, cov_ref = pj.task_posterior(P, X, Y, Xs)
    _, var_obs_ref = pj.observed_posterior(P, X, Y, Xs)
    m = m.to(DEV)
    m.eval()
    m.likelihood.eval()
    with torch.no_grad():
        full_likelihood = m.full_likelihood()
        dist = m(Xs.to(DEV))
        obs = full_likelihood(dist)                         # experiments.py:317-321
        assert dist.mean.shape == (ns, p)
        assert torch.allclose(dist.mean.cpu(), mean_ref, rtol=1e-8, atol=1e-10)
        assert torch.allclose(dist.variance.cpu(), torch.diagonal(cov_ref).reshape(ns, p), rtol=1e-7, atol=1e-10)
        assert torch.allclose(obs.variance.cpu(), var_obs_ref, rtol=1e-7, atol=1e-10)
        lo, hi = obs.confidence_region()
        assert torch.allclose((hi - lo).cpu(), 4 * var_obs_ref.sqrt(), rtol=1e-7)
        full = m(Xs.to(DEV), full_cov=True)
        assert torch.allclose(full.covariance_matrix.cpu(), cov_ref, rtol=1e-7, atol=1e-9)
        lat = m.compute_latent_distrib(Xs.to(DEV))
        assert lat.mean.shape == (q, ns)


def test_exact_gp_single_output_config1(plmc):
    """BASELINE config 1: ExactGPModel single-output RBF, n=512, d=4 (README.md:33-58 usage)."""
    n, d = 512, 4
    X, Y = _data(n, d, 1, seed=4)
    y = Y[:, 0].contiguous()
    lik = plmc.GaussianLikelihood()
    model = plmc.ExactGPModel(X, y, lik, mean_type=plmc.ConstantMean, kernel_type=plmc.RBFKernel).double()
    lik = lik.double()
    with torch.no_grad():
        model.mean_module.raw_constant.fill_(0.3)
        model.covar_module.raw_lengthscale.copy_(torch.tensor([[[0.2, -0.3, 0.5, 0.1]]], dtype=torch.float64))
    raw_ls = model.covar_module.raw_lengthscale.detach().clone().requires_grad_()
    raw_nz = lik.noise_covar.raw_noise.detach().clone().requires_grad_()
    c = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    ref = ld.exact_gp_mll("rbf", X, y, gm.softplus(raw_ls).reshape(1, d), gm.softplus(raw_nz).reshape(1) + 1e-4, c)
    ref.backward()

    model, lik = model.to(DEV), lik.to(DEV)
    model.train(); lik.train()
    mll = plmc.ExactMarginalLogLikelihood(lik, model)
    out = mll(model(X.to(DEV)), y.to(DEV))
    out.sum().backward()
    assert abs(float(out.sum()) - float(ref)) < 1e-10 * abs(float(ref))
    assert torch.allclose(model.covar_module.raw_lengthscale.grad.cpu(), raw_ls.grad, rtol=1e-6, atol=1e-9)
    assert torch.allclose(lik.noise_covar.raw_noise.grad.cpu(), raw_nz.grad, rtol=1e-6, atol=1e-9)
    assert torch.allclose(model.mean_module.raw_constant.grad.cpu(), c.grad, rtol=1e-6, atol=1e-9)
    # prediction
    Xs = 2 * torch.rand(50, d, dtype=torch.float64) - 1
    mu, cov = gm.exact_gp_posterior("rbf", X, gm.softplus(raw_ls.detach()).reshape(1, d),
                                    gm.softplus(raw_nz.detach()).reshape(1) + 1e-4, y[None], Xs, mean=torch.tensor([0.3], dtype=torch.float64))
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = model(Xs.to(DEV))
    assert torch.allclose(pred.mean.cpu(), mu[0], rtol=1e-8, atol=1e-10)
    assert torch.allclose(pred.variance.cpu(), torch.diagonal(cov[0]), rtol=1e-7, atol=1e-10)
    assert model.lscales().shape == (d,)


def test_fp32_projected_step_meets_the_1e4_target(plmc):
    """BASELINE target: fp32 log-likelihood within 1e-4 relative of the fp64 oracle."""
    n, d, p, q = 1024, 8, 8, 4
    X, Y = _data(n, d, p, seed=9)
    torch.manual_seed(0)
    m = _model(plmc, X.float(), Y.float(), q, plmc.MaternKernel, init_lmc_coeffs=True, **VARIANTS["PLMC_fast"])
    P = oracle_params(m)
    ref = float(pj.projected_mll(P, X, Y))
    m = m.to(DEV)
    m.train()
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    val = float(mll(m(X.float().to(DEV)), Y.float().to(DEV)))
    assert abs(val - ref) < 1e-4 * abs(ref), (val, ref)


@pytest.mark.parametrize("q,world", [(3, 2), (4, 4)])
def test_latent_shards_sum_to_the_unsharded_step(plmc, q, world):
    """The per-rank path of the multi-GPU run (latent_shard -> only this rank's latents go through the HIP engine,
    hyper-parameter gradient node included) on one GPU: the loss shares and the gradients of all `world` shards add
    up to the un-sharded loss and gradients (what the fused all-reduce of parallel.sync_loss_and_grads forms)."""
    n, d, p = 300, 3, 6
    X, Y = _data(n, d, p, seed=21)
    Xd, Yd = X.to(DEV), Y.to(DEV)

    def build(shard):
        torch.manual_seed(2)
        m = _model(plmc, X, Y, q, plmc.MaternKernel, init_lmc_coeffs=True, latent_shard=shard, **VARIANTS["PLMC"])
        m = perturb_(m.double()).to(DEV)
        m.train(); m.likelihood.train()
        return m, plmc.ProjectedLMCmll(m.likelihood, m)

    m0, mll0 = build(None)
    loss0 = -mll0(m0(Xd), Yd)
    loss0.backward()
    total, grads = 0.0, None
    for rank in range(world):
        m1, mll1 = build((rank, world))
        assert m1.latent_ids == list(range(rank, q, world))
        share = -mll1(m1(Xd), Yd)
        share.backward()
        total = total + float(share.detach())
        gs = [torch.zeros_like(prm) if prm.grad is None else prm.grad.clone() for prm in m1.parameters()]
        grads = gs if grads is None else [a + b for a, b in zip(grads, gs)]
    assert abs(total - float(loss0)) < 1e-10 * abs(float(loss0)), (total, float(loss0))
    for (name, prm), g in zip(m0.named_parameters(), grads):
        assert torch.allclose(prm.grad, g, rtol=1e-8, atol=1e-11), (name, (prm.grad - g).abs().max())


def test_sarcos_shaped_projected_model_p_equals_q(plmc):
    """SURVEY.md 8e's reading of BASELINE config 5 at a size the dense fp64 oracle still handles: a ProjectedGPModel with as many
    latents as tasks (p = q = 7: the p - q = 0 branches of projected_lmc.py:975-988, 1044-1047, 1210-1218), d = 21 inputs
    (the d > 8 instance of the gradient kernels), n = 3000 (three groups of block rows: the look-ahead schedule and the
    split engine), Matern-5/2.  Loss and every gradient against the oracle; the eval-mode posterior, and its per-rank
    partial sums over 7 latent shards (what the all-reduce of the 7-GPU run adds up), against the oracle's posterior."""
    n, d, p, q = 3000, 21, 7, 7
    X, Y = _data(n, d, p, seed=17)
    torch.manual_seed(3)
    m = _model(plmc, X, Y, q, plmc.MaternKernel, init_lmc_coeffs=True, **VARIANTS["PLMC"])
    m = perturb_(m.double(), scale=0.1)
    P = oracle_params(m)
    for k in pj.tensor_keys(P):
        P[k].requires_grad_(True)
    ref = -pj.projected_mll(P, X, Y)
    ref.backward()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    m.train(); m.likelihood.train()
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    loss = -mll(m(Xd), Yd)
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-9 * abs(float(ref)), (float(loss), float(ref))
    pm = param_map(m)
    for pname, prm in m.named_parameters():
        g_ref = P[pm[pname]].grad
        assert prm.grad is not None, pname
        assert torch.allclose(prm.grad.cpu(), g_ref, rtol=5e-6, atol=1e-8), (pname, (prm.grad.cpu() - g_ref).abs().max())
    # eval-mode posterior, un-sharded and as the sum of the 7 ranks' shares
    Xs = _data(257, d, p, seed=4)[0]
    with torch.no_grad():
        Pd = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in P.items()}
        mean_ref, cov_ref = pj.task_posterior(Pd, X, Y, Xs)
    var_ref = torch.diagonal(cov_ref).reshape(Xs.shape[0], p)
    m.eval()
    with torch.no_grad():
        out = m(Xs.to(DEV))
    assert (out.mean.cpu() - mean_ref).abs().max() < 1e-8 * max(1.0, float(mean_ref.abs().max()))
    assert (out.variance.cpu() - var_ref).abs().max() < 1e-8 * max(1.0, float(var_ref.abs().max()))
    from projectedlmc import _engine
    mean_sum, var_sum = torch.zeros_like(mean_ref), torch.zeros_like(var_ref)
    for rank in range(q):
        torch.manual_seed(3)
        ms = _model(plmc, X, Y, q, plmc.MaternKernel, init_lmc_coeffs=True, latent_shard=(rank, q), **VARIANTS["PLMC"]).double()
        ms.load_state_dict(sd)
        ms = ms.to(DEV).eval()
        with torch.no_grad():
            ml, vl = ms._latent_posterior(Xs.to(DEV))
            mean_r, var_r = _engine.mix_posterior(ml, vl, ms.lmc_coefficients().detach()[ms.latent_ids], 0.0)
        mean_sum += mean_r.cpu()
        var_sum += var_r.cpu()
    assert (mean_sum - mean_ref).abs().max() < 1e-8 * max(1.0, float(mean_ref.abs().max()))
    assert (var_sum + m.eps - var_ref).abs().max() < 1e-8 * max(1.0, float(var_ref.abs().max()))


def test_deferred_pivot_check_walks_the_same_jitter_ladder(plmc):
    """ProjectedLMCmll looks at the pivot check after the whole forward pass is queued and redoes the pass with jitter;
    the direct path (check inside the log-prob call) must land on the same jitter and the same loss."""
    n, d, p, q = 300, 2, 4, 2
    X, Y = _data(n, d, p, seed=3)
    torch.manual_seed(1)
    m = _model(plmc, X.float(), Y.float(), q, plmc.RBFKernel, noise_thresh=-40., **VARIANTS["PLMC_fast"])
    m = m.to(DEV)
    with torch.no_grad():                                   # long lengthscales: K is numerically singular in fp32
        for name, prm in m.named_parameters():
            if "lengthscale" in name:
                prm.fill_(5.0)
            if "raw_noise" in name:
                prm.fill_(-40.0)                            # noise ~ e^-40: nothing holds the spectrum up
    Xd, Yd = X.float().to(DEV), Y.float().to(DEV)
    m.train(); m.likelihood.train()
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    with pytest.warns(RuntimeWarning, match="not p.d."):      # in a training step the check may sit behind the backward pass
        out = mll(m(Xd), Yd)                                    # (settings.late_pivot_check: `out` is then overwritten with the
        loss = -out                                             # jittered value; tensors derived from it before that are stale)
        loss.backward()
    loss = -out.detach()
    assert torch.isfinite(loss)
    assert all(torch.isfinite(prm.grad).all() for prm in m.parameters() if prm.grad is not None)
    with pytest.warns(RuntimeWarning, match="not p.d."):
        direct = -mll._forward_once(m(Xd), Yd)
    assert abs(float(loss.detach()) - float(direct.detach())) <= 1e-5 * abs(float(direct.detach()))


def _singular_model(plmc, seed=1):
    n, d, p, q = 300, 2, 4, 2
    X, Y = _data(n, d, p, seed=3)
    torch.manual_seed(seed)
    m = _model(plmc, X.float(), Y.float(), q, plmc.RBFKernel, noise_thresh=-40., **VARIANTS["PLMC_fast"]).to(DEV)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if "lengthscale" in name:
                prm.fill_(5.0)
            if "raw_noise" in name:
                prm.fill_(-40.0)
    m.train(); m.likelihood.train()
    return m, X.float().to(DEV), Y.float().to(DEV)


def test_late_pivot_check_is_the_same_training_step(plmc):
    """settings.late_pivot_check (default on): the pivot check of a training step sits behind the backward pass.  Healthy
    matrices: losses and parameters of three AdamW steps are bit-identical to the check at the end of the forward pass.  A
    non-PD matrix: the gradients of the failed pass are taken back (gradients accumulated BEFORE the pass are kept), the jitter
    ladder redoes forward and backward inside backward(), and the loss tensor the caller holds ends up with the jittered value:
    same warnings, same loss, same gradients as with the early check."""
    from projectedlmc import settings

    def train(late):
        n, d, p, q = 300, 2, 4, 2
        X, Y = _data(n, d, p, seed=5)
        torch.manual_seed(2)
        m = _model(plmc, X.float(), Y.float(), q, plmc.MaternKernel, **VARIANTS["PLMC_fast"]).to(DEV)
        m.train(); m.likelihood.train()
        mll = plmc.ProjectedLMCmll(m.likelihood, m)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
        Xd, Yd = X.float().to(DEV), Y.float().to(DEV)
        losses = []
        with settings.late_pivot_check(late):
            for _ in range(3):
                opt.zero_grad()
                loss = -mll(m(Xd), Yd)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        return losses, [prm.detach().clone() for prm in m.parameters()]

    la, pa = train(True)
    lb, pb = train(False)
    assert la == lb
    assert all(torch.equal(a, b) for a, b in zip(pa, pb))

    def failing_step(late):
        m, Xd, Yd = _singular_model(plmc)
        mll = plmc.ProjectedLMCmll(m.likelihood, m)
        for prm in m.parameters():
            prm.grad = torch.full_like(prm, 0.25)               # gradients already accumulated by an earlier call
        with settings.late_pivot_check(late), warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            out = mll(m(Xd), Yd)
            n_fwd = sum("not p.d." in str(w.message) for w in rec)
            (-out).backward()
        n_all = sum("not p.d." in str(w.message) for w in rec)
        return -float(out.detach()), [prm.grad.clone() for prm in m.parameters()], n_fwd, n_all

    l1, g1, f1, a1 = failing_step(True)
    l0, g0, f0, a0 = failing_step(False)
    assert f1 == 0 and a1 >= 1                                  # the late check warns from inside backward()
    assert f0 == a0 == a1                                       # the same rungs of the ladder
    assert l1 == l0 and math.isfinite(l1)
    for a, b in zip(g1, g0):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_late_pivot_check_of_a_dropped_loss_is_settled_at_the_next_forward(plmc):
    """A loss evaluated with gradients on and never back-propagated: its pending check is looked at by the next forward call."""
    m, Xd, Yd = _singular_model(plmc)
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    _ = mll(m(Xd), Yd)                                          # no backward
    with pytest.warns(RuntimeWarning, match="never back-propagated"):
        with torch.no_grad():
            mll(m(Xd), Yd)
