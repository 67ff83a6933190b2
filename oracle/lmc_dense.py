"""Oracle: exact (dense) LMC / ICM multitask GP and the single/batch exact GP MLL.
TEST INFRASTRUCTURE ONLY.

Restates
* `MultitaskGPModel.forward` with `LCMKernel` / `MultitaskKernel`   projected_lmc.py:462-466,586-589
* `MultitaskGaussianLikelihood` + `ExactMarginalLogLikelihood`      experiments.py:184,233
* `ExactGPModel.forward` (+ batch of independent GPs)               projected_lmc.py:306-321

[gpytorch-knowledge] (unverified offline):
  IndexKernel  B = F F^T + diag(softplus(raw_var)),  F = covar_factor (p x rank)
  MultitaskKernel(x) = K_data(x,x) (x) B            (data-major: flat index = i_point*p + i_task)
  LCMKernel = sum_i MultitaskKernel_i, each rank 1
  MultitaskGaussianLikelihood(rank=0): Sigma = diag(softplus(raw_task_noises)+1e-4) + (softplus(raw_noise)+1e-4) I
                              (rank=r): Sigma = F_n F_n^T + noise I
  MultitaskMean(ConstantMean) -> n x p of per-task constants
  ExactMarginalLogLikelihood = MVN.log_prob(vec Y) / n   (n = number of points, :event_shape[0])
"""
import torch

from . import gp_math as gm


def task_covariances(covar_factor, raw_var):
    """covar_factor (q,p,rank), raw_var (q,p) -> B (q,p,p)."""
    return covar_factor @ covar_factor.transpose(-1, -2) + torch.diag_embed(gm.softplus(raw_var))


def task_noise_covariance(p, raw_task_noises=None, raw_noise=None, noise_factor=None, lb=1e-4, dtype=torch.float64):
    S = torch.zeros(p, p, dtype=dtype)
    if noise_factor is not None:
        S = S + noise_factor @ noise_factor.T
    if raw_task_noises is not None:
        S = S + torch.diag_embed(gm.softplus(raw_task_noises) + lb)
    if raw_noise is not None:
        S = S + (gm.softplus(raw_noise).reshape(()) + lb) * torch.eye(p, dtype=dtype)
    return S


def lmc_covariance(kind, X, ell, B, Sigma, nu=2.5, outputscale=None):
    """sum_i K_i (x) B_i + I_n (x) Sigma, data-major interleaved (np x np).
    ell (q,d) per-latent lengthscales (ICM: pass the same row q times or q=1 with B summed)."""
    n = X.shape[0]
    K = gm.kernel_matrix(kind, X, X, ell, outputscale, nu)
    C = torch.kron(torch.eye(n, dtype=X.dtype), Sigma)
    for i in range(K.shape[0]):
        C = C + torch.kron(K[i], B[i])
    return C


def lmc_exact_mll(kind, X, Y, ell, B, Sigma, mean_const=None, nu=2.5, outputscale=None):
    """ExactMarginalLogLikelihood of the exact LMC/ICM: log N(vec(Y - m); 0, C) / (n p)
    [gpytorch-knowledge, v1.11: num_data = function_dist.event_shape.numel() = n * p for a
    MultitaskMultivariateNormal -- the same expression the reference copies at projected_lmc.py:1194]."""
    n, p = Y.shape
    C = lmc_covariance(kind, X, ell, B, Sigma, nu, outputscale)
    R = Y if mean_const is None else Y - mean_const.reshape(1, p)
    return gm.mvn_log_prob(C, R.reshape(-1)) / (n * p)


def lmc_posterior(kind, X, Y, Xs, ell, B, Sigma, mean_const=None, nu=2.5, outputscale=None):
    """Posterior task mean (ns,p) and marginal variance (ns,p) of f (no observation noise)."""
    n, p = Y.shape
    ns = Xs.shape[0]
    C = lmc_covariance(kind, X, ell, B, Sigma, nu, outputscale)
    Ks = gm.kernel_matrix(kind, Xs, X, ell, outputscale, nu)
    Kss = gm.kernel_matrix(kind, Xs, Xs, ell, outputscale, nu)
    Cs = sum(torch.kron(Ks[i], B[i]) for i in range(Ks.shape[0]))           # (ns*p, n*p)
    Css = sum(torch.kron(Kss[i], B[i]) for i in range(Ks.shape[0]))
    R = Y if mean_const is None else Y - mean_const.reshape(1, p)
    L = torch.linalg.cholesky(C)
    alpha = torch.cholesky_solve(R.reshape(-1, 1), L)
    mu = (Cs @ alpha).reshape(ns, p)
    if mean_const is not None:
        mu = mu + mean_const.reshape(1, p)
    V = torch.linalg.solve_triangular(L, Cs.T, upper=False)
    var = (torch.diagonal(Css) - (V * V).sum(0)).reshape(ns, p)
    return mu, var


def exact_gp_mll(kind, X, Y, ell, noise, mean_const=None, nu=2.5, outputscale=None):
    """Batch of independent exact GPs (ExactGPModel with n_tasks=q; q=1 single output).
    Y: (n,) or (n,q).  Returns ExactMarginalLogLikelihood value:
    log_prob summed over the batch / n  [gpytorch-knowledge: for a batch MVN the MLL
    divides the (q,) log_prob by n; drivers then call .sum() or pass a
    MultitaskMultivariateNormal whose log_prob already sums over tasks]."""
    n = X.shape[0]
    Yt = Y.reshape(n, -1).T                                                  # (q,n)
    if mean_const is not None:
        Yt = Yt - mean_const.reshape(-1, 1)
    lp = gm.exact_latent_log_prob(kind, X, ell, noise, Yt, outputscale, nu)
    return lp.sum() / n
