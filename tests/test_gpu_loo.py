"""GPU parity of `compute_loo` (projected_lmc.py:371-436, 1108-1119): the
leave-one-out variances 1/diag(Khat^-1) and residuals Khat^-1 y * sigma2 share the blocked sweep."""
import warnings

import pytest
import torch

from oracle import gp_math as gm
from oracle import projected as pj
from _bridge import oracle_params, perturb_

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _loo_ref(kind, nu, X, ell, noise, y):
    K = gm.kernel_matrix(kind, X, X, ell, None, nu) + noise.reshape(-1, 1, 1) * torch.eye(X.shape[0], dtype=X.dtype)
    Kinv = torch.cholesky_inverse(torch.linalg.cholesky(K))      # SPD: no LU (batched MKL getri is flaky after big factorisations)
    s2 = 1.0 / torch.diagonal(Kinv, dim1=-2, dim2=-1)
    return s2, (Kinv @ y.unsqueeze(-1)).squeeze(-1) * s2


def test_projected_model_loo():
    import projectedlmc as plmc
    g = torch.Generator().manual_seed(0)
    n, d, p, q = 260, 3, 5, 2
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(n, p, generator=g, dtype=torch.float64)
    torch.manual_seed(2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = plmc.ProjectedGPModel(X, Y, p, q, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                  init_lmc_coeffs=True, BDN=False)
    m = perturb_(m.double())
    P = oracle_params(m)
    s2_ref, r_ref = _loo_ref("matern", 2.5, X, pj.lengthscale(P), pj.projected_noise(P), pj.project_data(P, Y))
    m = m.to(DEV)
    s2, r = m.compute_loo()
    assert s2.shape == (n, q)
    assert torch.allclose(s2.cpu(), s2_ref.T, rtol=1e-8)
    assert torch.allclose(r.cpu(), r_ref.T, rtol=1e-7, atol=1e-10)


def test_exact_gp_loo_single_output():
    import projectedlmc as plmc
    g = torch.Generator().manual_seed(1)
    n, d = 300, 2
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    y = torch.randn(n, generator=g, dtype=torch.float64)
    lik = plmc.GaussianLikelihood().double()
    model = plmc.ExactGPModel(X, y, lik, mean_type=plmc.ZeroMean, kernel_type=plmc.RBFKernel).double()
    ell = model.covar_module.lengthscale.detach().reshape(1, d)
    noise = lik.noise.detach().reshape(1)
    s2_ref, r_ref = _loo_ref("rbf", 2.5, X, ell, noise, y[None])
    model = model.to(DEV)
    s2, r = model.compute_loo()
    assert s2.shape == (n,)
    assert torch.allclose(s2.cpu(), s2_ref[0], rtol=1e-8)
    assert torch.allclose(r.cpu(), r_ref[0], rtol=1e-7, atol=1e-10)
