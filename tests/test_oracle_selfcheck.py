"""Reference-independent checks that pin the oracle: closed forms,
finite differences, the projected == dense-LMC identity, batch == loop, p == q edge."""
import math

import pytest
import torch

from oracle import gp_math as gm
from oracle import projected as pj
from oracle import lmc_dense as ld



@pytest.fixture(autouse=True)
def _float64_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


def _data(n=40, d=3, p=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g) - 1
    Y = torch.randn(n, p, generator=g)
    return X, Y


def test_kernel_closed_forms():
    X1 = torch.tensor([[0.0, 0.0]])
    X2 = torch.tensor([[0.3, -0.4]])
    ell = torch.tensor([[0.5, 2.0]])
    r2 = (0.3 / 0.5) ** 2 + (0.4 / 2.0) ** 2
    r = math.sqrt(r2)
    assert torch.allclose(gm.kernel_matrix("rbf", X1, X2, ell)[0, 0, 0], torch.tensor(math.exp(-0.5 * r2)))
    m52 = (1 + math.sqrt(5) * r + 5 * r2 / 3) * math.exp(-math.sqrt(5) * r)
    assert torch.allclose(gm.kernel_matrix("matern", X1, X2, ell, nu=2.5)[0, 0, 0], torch.tensor(m52))
    m32 = (1 + math.sqrt(3) * r) * math.exp(-math.sqrt(3) * r)
    assert torch.allclose(gm.kernel_matrix("matern", X1, X2, ell, nu=1.5)[0, 0, 0], torch.tensor(m32))
    assert torch.allclose(gm.kernel_matrix("matern", X1, X2, ell, nu=0.5)[0, 0, 0], torch.tensor(math.exp(-r)))
    # chunked builder agrees
    X, _ = _data()
    e = torch.tensor([0.7, 0.4, 1.1])
    for kind in ("rbf", "matern"):
        a = gm.kernel_matrix(kind, X, X, e[None])[0]
        b = gm.kernel_matrix_chunked(kind, X, X, e, chunk=16)
        assert torch.allclose(a, b, atol=1e-12)


def test_mvn_log_prob_matches_torch_distribution():
    X, Y = _data()
    K = gm.kernel_matrix("rbf", X, X, torch.tensor([[0.5, 0.6, 0.7]]))[0] + 0.1 * torch.eye(40)
    ref = torch.distributions.MultivariateNormal(torch.zeros(40), K).log_prob(Y[:, 0])
    assert torch.allclose(gm.mvn_log_prob(K, Y[:, 0]), ref)


@pytest.mark.parametrize("kind,nu,oscale", [("rbf", 2.5, False), ("matern", 2.5, True), ("matern", 1.5, False),
                                            ("matern", 0.5, False)])
def test_analytic_gradient_matches_autograd_and_fd(kind, nu, oscale):
    X, Y = _data(n=30, d=3, p=2)
    q = 2
    ell = (0.4 + 0.3 * torch.rand(q, 3)).requires_grad_()
    noise = torch.tensor([0.05, 0.3], requires_grad=True)
    osc = torch.tensor([1.3, 0.7], requires_grad=True) if oscale else None
    ytil = Y.T.clone().requires_grad_()
    lp = gm.exact_latent_log_prob(kind, X, ell, noise, ytil, osc, nu)
    ins = [ell, noise, ytil] + ([osc] if oscale else [])
    grads = torch.autograd.grad(lp.sum(), ins)
    lp2, g_ell, g_noise, g_os, g_y = gm.exact_latent_log_prob_analytic(kind, X, ell.detach(), noise.detach(),
                                                                     ytil.detach(), None if osc is None else osc.detach(), nu)
    assert torch.allclose(lp, lp2, rtol=1e-11)
    assert torch.allclose(grads[0], g_ell, rtol=1e-7, atol=1e-9)
    assert torch.allclose(grads[1], g_noise, rtol=1e-7, atol=1e-9)
    assert torch.allclose(grads[2], g_y, rtol=1e-7, atol=1e-9)
    if oscale:
        assert torch.allclose(grads[3], g_os, rtol=1e-7, atol=1e-9)
    # central finite difference on one lengthscale entry
    h = 1e-6
    e1, e2 = ell.detach().clone(), ell.detach().clone()
    e1[1, 2] += h
    e2[1, 2] -= h
    f = lambda e: gm.exact_latent_log_prob(kind, X, e, noise.detach(), ytil.detach(),
                                           None if osc is None else osc.detach(), nu).sum()
    fd = (f(e1) - f(e2)) / (2 * h)
    assert abs(fd - g_ell[1, 2]) < 1e-5 * max(1.0, abs(fd))


VARIANTS = {
    "PLMC": dict(BDN=False, diagonal_B=False, scalar_B=False),          # experiments.py:197-200
    "PLMC_diagB": dict(BDN=False, diagonal_B=True, scalar_B=False),
    "BDN_fullB": dict(BDN=True, diagonal_B=False, scalar_B=False),
    "BDN_diagB": dict(BDN=True, diagonal_B=True, scalar_B=False),
    "PLMC_fast": dict(BDN=True, diagonal_B=True, scalar_B=True),        # experiments.py:212-215
}


def _perturb(P, seed=1):
    g = torch.Generator().manual_seed(seed)
    for k in pj.tensor_keys(P):
        P[k] = P[k] + 0.3 * torch.randn(P[k].shape, generator=g)
    if "B_tilde_inv_chol_raw" in P:
        P["B_tilde_inv_chol_raw"] = P["B_tilde_inv_chol_raw"].tril()
    return P


@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("kind", ["rbf", "matern"])
@pytest.mark.parametrize("init", [False, True])
def test_projected_mll_equals_dense_lmc_density(name, kind, init):
    """The paper's central identity, embodied by projected_lmc.py:1023-1060 vs :1199-1240."""
    X, Y = _data(n=40, d=2, p=5)
    fc = torch.randn(5, 2, generator=torch.Generator().manual_seed(3))
    P = pj.init_params(X, Y, 2, kind=kind, init_lmc_coeffs=init, fake_coeffs=fc, outputscales=(kind == "rbf"),
                       **VARIANTS[name])
    P = _perturb(P)
    mll = pj.projected_mll(P, X, Y)
    dense = pj.dense_lmc_log_density(P, X, Y)
    assert torch.allclose(mll * X.shape[0], dense, rtol=1e-9), (float(mll * 40), float(dense))


NONBULK = {
    # realdata_experiments.py:107-111: the OILMM configuration (orthogonal Q, diagonal R, scalar B, block-diagonal noise)
    "oilmm": dict(BDN=True, diagonal_B=True, scalar_B=True, diagonal_R=True),
    "PLMC": dict(BDN=False, diagonal_B=False, scalar_B=False),
    "BDN_diagB_cayley": dict(BDN=True, diagonal_B=True, scalar_B=False, ortho_param="cayley"),
    "PLMC_householder": dict(BDN=False, diagonal_B=True, scalar_B=False, ortho_param="householder"),
}


@pytest.mark.parametrize("name", list(NONBULK))
@pytest.mark.parametrize("init", [False, True])
def test_projected_mll_equals_dense_lmc_density_with_separately_parametrised_Q_and_R(name, init):
    """The same identity for `bulk=False` (projected_lmc.py:851-853, :873, :884, :963-970 and the loss branch :1237): Q_plus
    through torch's orthogonal parametrisation, R diagonal-positive or upper-triangular with exp on the diagonal."""
    X, Y = _data(n=40, d=2, p=5)
    fc = torch.randn(5, 2, generator=torch.Generator().manual_seed(3))
    P = pj.init_params(X, Y, 2, kind="matern", init_lmc_coeffs=init, fake_coeffs=fc, bulk=False, **NONBULK[name])
    # at initialisation the parametrised pair reproduces the bulk matrix H = Q R of the same draw
    Pb = pj.init_params(X, Y, 2, kind="matern", init_lmc_coeffs=init, fake_coeffs=fc,
                        **{k: v for k, v in NONBULK[name].items() if k not in ("diagonal_R", "ortho_param")})
    assert torch.allclose(pj.lmc_coefficients(P), pj.lmc_coefficients(Pb), atol=1e-12)
    assert torch.allclose(pj.projected_mll(P, X, Y), pj.projected_mll(Pb, X, Y), rtol=1e-10)
    P = _perturb(P)
    if P["ortho_param"] == "householder":                      # the signed diagonal is not a free parameter
        P["Q_plus_original"].diagonal().copy_(torch.sign(P["Q_plus_original"].diagonal()))
    Q, R, Q_orth = pj.QR(P)
    assert torch.allclose(Q.T @ Q, torch.eye(2), atol=1e-12)
    if Q_orth.numel():
        assert torch.allclose(Q.T @ Q_orth, torch.zeros(2, 3), atol=1e-12)
    mll = pj.projected_mll(P, X, Y)
    dense = pj.dense_lmc_log_density(P, X, Y)
    assert torch.allclose(mll * X.shape[0], dense, rtol=1e-9), (float(mll * 40), float(dense))


@pytest.mark.parametrize("name", ["PLMC", "PLMC_fast", "BDN_diagB"])
def test_p_equals_q_edge(name):
    X, Y = _data(n=30, d=2, p=3)
    fc = torch.randn(3, 3, generator=torch.Generator().manual_seed(5))
    P = pj.init_params(X, Y, 3, fake_coeffs=fc, **VARIANTS[name])
    P = _perturb(P, seed=2)
    mll = pj.projected_mll(P, X, Y)
    dense = pj.dense_lmc_log_density(P, X, Y)
    assert torch.allclose(mll * 30, dense, rtol=1e-9)


def test_projection_matrix_consistent_with_project_data():
    X, Y = _data()
    for name, kw in VARIANTS.items():
        P = _perturb(pj.init_params(X, Y, 2, **kw))
        T = pj.projection_matrix(P)
        assert torch.allclose((Y @ T).T, pj.project_data(P, Y), atol=1e-10), name
        # T^T H = I_q  (T is a generalised inverse of the mixing matrix)
        Ht = pj.lmc_coefficients(P)
        assert torch.allclose(T.T @ Ht.T, torch.eye(2), atol=1e-9), name


def test_batch_equals_loop():
    X, Y = _data(n=25, d=3, p=4)
    ell = 0.3 + torch.rand(4, 3)
    noise = 0.05 + torch.rand(4)
    lp = gm.exact_latent_log_prob("matern", X, ell, noise, Y.T)
    for i in range(4):
        one = gm.exact_latent_log_prob("matern", X, ell[i:i + 1], noise[i:i + 1], Y.T[i:i + 1])
        assert torch.allclose(lp[i], one[0], rtol=1e-12)


def test_task_posterior_matches_dense_lmc_posterior():
    """Eval-mode Kronecker recombination (projected_lmc.py:1140-1153) == conditioning the dense
    LMC prior with Sigma = full_noise_covariance."""
    X, Y = _data(n=30, d=2, p=4)
    Xs = 2 * torch.rand(7, 2) - 1
    P = _perturb(pj.init_params(X, Y, 2, kind="matern", **VARIANTS["PLMC"]))
    mean, cov = pj.task_posterior(P, X, Y, Xs)
    Ht = pj.lmc_coefficients(P)
    B = torch.stack([torch.outer(Ht[i], Ht[i]) for i in range(2)])
    mu, var = ld.lmc_posterior("matern", X, Y, Xs, pj.lengthscale(P), B, pj.full_noise_covariance(P))
    assert torch.allclose(mean, mu, atol=1e-8)
    assert torch.allclose(torch.diagonal(cov).reshape(7, 4) - P["eps"], var, atol=1e-8)


def test_lmc_exact_mll_rank1_equals_projected_dense():
    X, Y = _data(n=20, d=2, p=3)
    q = 2
    ell = 0.4 + torch.rand(q, 2)
    F = torch.randn(q, 3, 1)
    raw_var = torch.randn(q, 3)
    B = ld.task_covariances(F, raw_var)
    S = ld.task_noise_covariance(3, raw_task_noises=torch.zeros(3), raw_noise=torch.zeros(1))
    v = ld.lmc_exact_mll("rbf", X, Y, ell, B, S)
    C = ld.lmc_covariance("rbf", X, ell, B, S)
    ref = torch.distributions.MultivariateNormal(torch.zeros(60), C).log_prob(Y.reshape(-1)) / 60
    assert torch.allclose(v, ref)
    # interleaving: entry ((a,s),(b,t)) = sum_i K_i[a,b] B_i[s,t] + delta_ab Sigma[s,t]
    K = gm.kernel_matrix("rbf", X, X, ell)
    a, s, b, t = 3, 1, 7, 2
    assert torch.allclose(C[a * 3 + s, b * 3 + t], (K[:, a, b] * B[:, s, t]).sum())


@pytest.mark.parametrize("ortho_param", ["matrix_exp", "cayley", "householder"])
@pytest.mark.parametrize("shape", [(6, 6), (6, 2)])
def test_restated_orthogonal_map_equals_torchs_parametrisation(ortho_param, shape):
    """`oracle.projected.orthogonal_map` restates what torch.nn.utils.parametrizations.orthogonal evaluates (the third-party
    arithmetic behind `bulk=False`, projected_lmc.py:963-965): element for element against torch's own module on perturbed
    parameters, square (mode 'Q_plus') and tall (mode 'Q') matrices."""
    torch.manual_seed(2)
    lin = torch.nn.Linear(shape[1], shape[0], bias=False).double()            # weight: shape[0] x shape[1]
    Q0, _ = torch.linalg.qr(torch.randn(shape[0], shape[0]))
    with torch.no_grad():
        lin.weight.copy_(Q0[:, :shape[1]])
    lin = torch.nn.utils.parametrizations.orthogonal(lin, name="weight", orthogonal_map=ortho_param,
                                                     use_trivialization=(ortho_param != "householder"))
    orth = lin.parametrizations.weight[0]
    with torch.no_grad():
        orig = lin.parametrizations.weight.original
        pert = 0.3 * torch.randn(orig.shape)
        if ortho_param == "householder":
            pert = pert.tril(-1)                                                # the signed diagonal is not a free parameter
        orig.add_(pert)
    base = getattr(orth, "base", None)
    got = pj.orthogonal_map(lin.parametrizations.weight.original.detach(), None if base is None else base.detach(), ortho_param)
    assert torch.allclose(got, lin.weight.detach(), atol=1e-13)
    assert torch.allclose(got.T @ got, torch.eye(shape[1]), atol=1e-12)


def test_prediction_cache_setting_rejects_unknown_modes():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "projected-lmc_amd"))
    from projectedlmc import settings
    assert settings.prediction_cache.value() == "lazy"
    with settings.prediction_cache("off"):
        assert settings.prediction_cache.value() == "off"
    assert settings.prediction_cache.value() == "lazy"
    with pytest.raises(ValueError):
        settings.prediction_cache("sometimes")
