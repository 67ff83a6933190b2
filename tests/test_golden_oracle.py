"""The committed golden vectors (tests/golden/projected_v1.json, made by tests/golden/make_golden.py) against
the oracle: guards the CPU restatement against drift.  The GPU counterpart is tests/test_gpu_golden.py."""
import pytest
import torch

from oracle import projected as pj
from oracle import gp_math as gm
from _bridge import oracle_params, param_map
from _golden import cases, T, build_projected


@pytest.fixture(autouse=True)
def _f64():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)


@pytest.mark.parametrize("c", cases("projected"), ids=lambda c: c["name"])
def test_projected_cases_reproduce(c):
    import projectedlmc as plmc
    m, X, Y = build_projected(plmc, c)
    P = oracle_params(m)
    for k in pj.tensor_keys(P):
        P[k].requires_grad_(True)
    loss = -pj.projected_mll(P, X, Y)
    loss.backward()
    assert abs(float(loss.detach()) - c["loss"]) <= 1e-11 * abs(c["loss"])
    pm = param_map(m)
    for pname, g in c["grads"].items():
        assert torch.allclose(P[pm[pname]].grad.reshape(-1), T(g).reshape(-1), rtol=1e-8, atol=1e-12), pname
    with torch.no_grad():
        mean, cov = pj.task_posterior(P, X, Y, T(c["Xs"]))
    assert torch.allclose(mean, T(c["pred_mean"]), rtol=1e-9, atol=1e-12)
    assert torch.allclose(torch.diagonal(cov).reshape(mean.shape), T(c["pred_var"]), rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("c", cases("exact"), ids=lambda c: c["name"])
def test_exact_cases_reproduce(c):
    kind = "rbf" if c["kernel"] == "RBFKernel" else "matern"
    X, y = T(c["X"]), T(c["y"])
    ell = gm.softplus(T(c["raw_lengthscale"]))
    noise = gm.softplus(T(c["raw_noise"])) + c["noise_lower_bound"]
    mll = gm.exact_latent_log_prob(kind, X, ell, noise, y[None], None, 2.5)[0] / X.shape[0]
    assert abs(float(mll) - c["mll"]) <= 1e-11 * abs(c["mll"])


def test_svd_init_case():
    """init_lmc_coefficients (projected_lmc.py:183-201): product (eigendecomposition, svd_flip signs) and oracle
    (SVD) agree with the stored coefficients."""
    import projectedlmc as plmc
    (c,) = cases("svd")
    Y = T(c["Y"])
    assert torch.allclose(pj.svd_init(Y, c["n_latents"]), T(c["coeffs"]), rtol=1e-9, atol=1e-12)
    assert torch.allclose(plmc.init_lmc_coefficients(Y, c["n_latents"]).to(torch.float64), T(c["coeffs"]), rtol=1e-7, atol=1e-9)
