"""Oracle: kernels, constraints and the dense exact-GP marginal log-likelihood.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, in plain torch-CPU:

* kernel construction of `handle_covar_`           projected_lmc.py:107-181
  (ARD kernel with batch_shape=[n_funcs], optional ScaleKernel)          :151-167
* `ExactGPModel.forward`  -> MVN(mean, K)                                 :306-321
* `likelihood(dist)` / `log_prob`  (K + sigma^2 I, dense Cholesky)        :1200-1201
* `gp.mlls.ExactMarginalLogLikelihood`                 experiments.py:233, README.md:45

[gpytorch-knowledge] (gpytorch==1.11, unverified offline):
  Positive() constraint = softplus, raw parameters initialised to 0;
  GreaterThan(lb) = softplus(raw) + lb;
  RBF   k = exp(-1/2 * sum_k ((x_k - x'_k)/l_k)^2)
  Matern k = exp(-sqrt(2 nu) r) * {1, 1 + sqrt3 r, 1 + sqrt5 r + 5/3 r^2},  r = ||(x-x')/l||
  ScaleKernel k = softplus(raw_outputscale) * k_base
  MVN.log_prob(y) = -1/2 (y-m)^T K^-1 (y-m) - 1/2 logdet K - n/2 log 2pi
  ExactMarginalLogLikelihood = log_prob / n  (priors omitted: none in scope).
"""
import math
import torch

LOG2PI = math.log(2.0 * math.pi)


def softplus(x):
    return torch.nn.functional.softplus(x)


def inv_softplus(y):
    y = torch.as_tensor(y)
    return y + torch.log(-torch.expm1(-y))


def kernel_matrix(kind, X1, X2, ell, outputscale=None, nu=2.5):
    """Batched ARD covariance.  X1: (n1,d), X2: (n2,d), ell: (q,d) -> (q,n1,n2).

    kind in {"rbf", "matern", "spline"}.  Uses direct differences (x-x')/l, which equals
    gpytorch's mean-centred |a|^2+|b|^2-2ab form in exact arithmetic.  "spline" restates the reference's own
    SplineKernel.forward (projected_lmc.py:26-36; `ell` only gives the batch size).
    """
    if kind == "spline":
        mins = torch.min(X1.unsqueeze(-2), X2.unsqueeze(-3))
        maxes = torch.max(X1.unsqueeze(-2), X2.unsqueeze(-3))
        K = (1 + mins * maxes + 0.5 * mins ** 2 * (maxes - mins / 3)).prod(dim=-1)
        K = K.unsqueeze(0).expand(ell.shape[0], -1, -1)
        return K if outputscale is None else K * outputscale.reshape(-1, 1, 1)
    ell = ell.reshape(ell.shape[0], 1, -1)                      # (q,1,d)
    a = X1.unsqueeze(0) / ell                                   # (q,n1,d)
    b = X2.unsqueeze(0) / ell
    diff = a.unsqueeze(2) - b.unsqueeze(1)                      # (q,n1,n2,d)
    r2 = (diff * diff).sum(-1)
    if kind == "rbf":
        K = torch.exp(-0.5 * r2)
    elif kind == "matern":
        r = torch.sqrt(r2.clamp_min(1e-30))
        e = torch.exp(-math.sqrt(2.0 * nu) * r)
        if nu == 0.5:
            K = e
        elif nu == 1.5:
            K = (1.0 + math.sqrt(3.0) * r) * e
        elif nu == 2.5:
            K = (1.0 + math.sqrt(5.0) * r + (5.0 / 3.0) * r2) * e
        else:
            raise ValueError("nu must be 0.5, 1.5 or 2.5")
    else:
        raise ValueError("unknown kernel kind %r" % (kind,))
    if outputscale is not None:
        K = K * outputscale.reshape(-1, 1, 1)
    return K


def kernel_matrix_chunked(kind, X1, X2, ell_1d, outputscale=None, nu=2.5, chunk=1024):
    """Single-latent kernel matrix built in row chunks (bounded memory; used for the
    n=8192 CPU baseline and full-size checks).  ell_1d: (d,) -> (n1,n2)."""
    rows = []
    for s in range(0, X1.shape[0], chunk):
        a = X1[s:s + chunk] / ell_1d
        b = X2 / ell_1d
        r2 = (a * a).sum(-1, keepdim=True) + (b * b).sum(-1)[None, :] - 2.0 * a @ b.T
        r2 = r2.clamp_min(0.0)
        if kind == "rbf":
            Kc = torch.exp(-0.5 * r2)
        else:
            r = torch.sqrt(r2 + 1e-30)
            e = torch.exp(-math.sqrt(2.0 * nu) * r)
            if nu == 0.5:
                Kc = e
            elif nu == 1.5:
                Kc = (1.0 + math.sqrt(3.0) * r) * e
            else:
                Kc = (1.0 + math.sqrt(5.0) * r + (5.0 / 3.0) * r2) * e
        rows.append(Kc)
    K = torch.cat(rows, 0)
    if outputscale is not None:
        K = K * outputscale
    return K


def mvn_log_prob(K, y, mean=None):
    """log N(y; mean, K) by dense Cholesky.  K: (...,n,n), y: (...,n) -> (...)."""
    if mean is not None:
        y = y - mean
    L = torch.linalg.cholesky(K)
    z = torch.linalg.solve_triangular(L, y.unsqueeze(-1), upper=False).squeeze(-1)
    quad = (z * z).sum(-1)
    logdet = 2.0 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
    n = y.shape[-1]
    return -0.5 * (quad + logdet + n * LOG2PI)


def exact_latent_log_prob(kind, X, ell, noise, ytil, outputscale=None, nu=2.5):
    """Per-latent log N(ytil_i; 0, K_i + noise_i I).  ell (q,d), noise (q,), ytil (q,n) -> (q,).
    Restates projected_lmc.py:1200-1201 for a zero-mean batch of q GPs."""
    K = kernel_matrix(kind, X, X, ell, outputscale, nu)
    n = X.shape[0]
    K = K + noise.reshape(-1, 1, 1) * torch.eye(n, dtype=K.dtype)
    return mvn_log_prob(K, ytil)


def exact_latent_log_prob_analytic(kind, X, ell, noise, ytil, outputscale=None, nu=2.5):
    """Same value plus the closed-form gradient the HIP path implements
    (SURVEY.md §8a row a4): dL/dtheta = 1/2 tr((aa^T - K^-1) dK/dtheta), dL/dy = -a.
    Returns (logp (q,), g_ell (q,d), g_noise (q,), g_outputscale (q,)|None, g_y (q,n))."""
    q, d = ell.shape
    n = X.shape[0]
    s = outputscale if outputscale is not None else torch.ones(q, dtype=X.dtype)
    Kb = kernel_matrix(kind, X, X, ell, None, nu)                      # unscaled
    K = Kb * s.reshape(-1, 1, 1) + noise.reshape(-1, 1, 1) * torch.eye(n, dtype=X.dtype)
    L = torch.linalg.cholesky(K)
    Kinv = torch.cholesky_inverse(L)
    alpha = (Kinv @ ytil.unsqueeze(-1)).squeeze(-1)
    logdet = 2.0 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
    logp = -0.5 * ((alpha * ytil).sum(-1) + logdet + n * LOG2PI)
    W = alpha.unsqueeze(-1) * alpha.unsqueeze(-2) - Kinv               # (q,n,n)
    diff = X.unsqueeze(1) - X.unsqueeze(0)                             # (n,n,d)
    d2 = diff * diff
    r2 = (d2.unsqueeze(0) / (ell * ell).reshape(q, 1, 1, d)).sum(-1)
    if kind == "rbf":
        base = Kb                                                      # dK/dl_k = K * D_k^2 / l_k^3
    else:
        r = torch.sqrt(r2.clamp_min(1e-30))
        e = torch.exp(-math.sqrt(2.0 * nu) * r)
        if nu == 2.5:
            base = (5.0 / 3.0) * (1.0 + math.sqrt(5.0) * r) * e
        elif nu == 1.5:
            base = 3.0 * e
        else:
            base = e / r.clamp_min(1e-15)
    WB = W * base * s.reshape(-1, 1, 1)
    g_ell = 0.5 * torch.einsum("qab,abk->qk", WB, d2) / ell ** 3
    g_noise = 0.5 * torch.diagonal(W, dim1=-2, dim2=-1).sum(-1)
    g_os = 0.5 * (W * Kb).sum((-2, -1)) if outputscale is not None else None
    return logp, g_ell, g_noise, g_os, -alpha


def exact_gp_posterior(kind, X, ell, noise, ytil, Xs, outputscale=None, nu=2.5, mean=None):
    """Posterior of a zero-mean (or constant-mean) batch of q GPs at Xs.
    [gpytorch-knowledge] DefaultPredictionStrategy: mu* = m + K* a, S* = K** - K* Khat^-1 K*^T.
    Returns (mean (q,ns), covar (q,ns,ns))."""
    n = X.shape[0]
    K = kernel_matrix(kind, X, X, ell, outputscale, nu) + noise.reshape(-1, 1, 1) * torch.eye(n, dtype=X.dtype)
    Ks = kernel_matrix(kind, Xs, X, ell, outputscale, nu)
    Kss = kernel_matrix(kind, Xs, Xs, ell, outputscale, nu)
    L = torch.linalg.cholesky(K)
    resid = ytil if mean is None else ytil - mean.reshape(-1, 1)
    alpha = torch.cholesky_solve(resid.unsqueeze(-1), L).squeeze(-1)
    mu = (Ks @ alpha.unsqueeze(-1)).squeeze(-1)
    if mean is not None:
        mu = mu + mean.reshape(-1, 1)
    V = torch.linalg.solve_triangular(L, Ks.transpose(-1, -2), upper=False)
    cov = Kss - V.transpose(-1, -2) @ V
    return mu, cov
