"""Oracle: the variational LMC (SVGP) path.  TEST INFRASTRUCTURE ONLY.

Restates `VariationalMultitaskGPModel` + `CustomLMCVariationalStrategy` (projected_lmc.py:659-813)
and `gp.mlls.VariationalELBO` (experiments.py:236) in plain torch-CPU.

[gpytorch-knowledge] (gpytorch==1.11, unverified offline):
  whitened VariationalStrategy, per latent i (inducing points Z shared, m x d):
     L L^T = K_ZZ + jit I,   jit = variational_cholesky_jitter (1e-4 fp32 / 1e-6 fp64)
     A = L^-1 K_ZX                               (m x n)
     mean_f = A^T m_i                            (m_i = variational_mean, init 0)
     cov_f  = K_XX + jit I + A^T (S_i - I) A     (S_i = Ls_i Ls_i^T, chol_variational_covar init I)
     KL_i   = 1/2 (tr S_i + m_i^T m_i - m - log det S_i)          (prior N(0, I))
  LMCVariationalStrategy: task mean = mean_f^T H (n x p), task covariance = sum_i cov_f,i (x) h_i h_i^T;
  CustomLMCVariationalStrategy adds the task-level means (:682-683).
  Gaussian expected_log_prob uses only marginal variances and the DIAGONAL of the task-noise
  covariance:  -1/2 sum_t [ ((y-mu)^2 + var) / s_t + log s_t + log 2pi ].
  VariationalELBO = (1/n) sum_points E_q[log p(y|f)]  -  KL / num_data.
  UnwhitenedVariationalStrategy (selected at projected_lmc.py:724-729 when train_ind_ratio == 1, Z = train_x):
     prior p(u) = N(0, Khat), Khat = K_ZZ + 1e-3 I (`add_jitter()` default);  q(u) = N(m, S);
     x == Z: q(f) = q(u);  otherwise mean = K_xZ Khat^-1 m, cov = K_xx - K_xZ Khat^-1 K_Zx + K_xZ Khat^-1 S Khat^-1 K_Zx;
     KL = KL(q(u) || p(u)) (closed form);  first call initialises q(u) to the prior.
"""
import math

import torch

from . import gp_math as gm


def latent_predictive(kind, X, Z, ell, var_mean, chol_var, nu=2.5, outputscale=None, jitter=1e-6):
    """(mean_f (q,n), var_f (q,n), KL (q,)) of the q whitened SVGPs."""
    m = Z.shape[0]
    eye = torch.eye(m, dtype=X.dtype)
    Kzz = gm.kernel_matrix(kind, Z, Z, ell, outputscale, nu) + jitter * eye
    Kzx = gm.kernel_matrix(kind, Z, X, ell, outputscale, nu)
    L = torch.linalg.cholesky(Kzz)
    A = torch.linalg.solve_triangular(L, Kzx, upper=False)                     # (q,m,n)
    mean_f = (A.transpose(-1, -2) @ var_mean.unsqueeze(-1)).squeeze(-1)
    Ls = chol_var.tril()
    Bm = Ls.transpose(-1, -2) @ A
    os_ = torch.ones(ell.shape[0], dtype=X.dtype) if outputscale is None else outputscale
    var_f = os_[:, None] + jitter - (A * A).sum(-2) + (Bm * Bm).sum(-2)
    S_tr = (Ls * Ls).sum((-2, -1))
    logdetS = 2.0 * torch.log(torch.diagonal(Ls, dim1=-2, dim2=-1).abs()).sum(-1)
    kl = 0.5 * (S_tr + (var_mean * var_mean).sum(-1) - m - logdetS)
    return mean_f, var_f, kl


def unwhitened_latent_predictive(kind, X, Z, ell, var_mean, chol_var, nu=2.5, outputscale=None, jitter=1e-3):
    """(mean_f (q,n), var_f (q,n), KL (q,)) of the q un-whitened variational GPs (see the module docstring)."""
    m = Z.shape[0]
    Khat = gm.kernel_matrix(kind, Z, Z, ell, outputscale, nu) + jitter * torch.eye(m, dtype=X.dtype)
    Ls = chol_var.tril()
    S = Ls @ Ls.transpose(-1, -2)
    Kinv = torch.linalg.inv(Khat)
    logdetK = torch.linalg.slogdet(Khat)[1]
    logdetS = 2.0 * torch.log(torch.diagonal(Ls, dim1=-2, dim2=-1).abs()).sum(-1)
    quad = (var_mean.unsqueeze(-2) @ Kinv @ var_mean.unsqueeze(-1)).reshape(-1)
    kl = 0.5 * ((Kinv * S).sum((-2, -1)) + quad - m + logdetK - logdetS)
    if X.shape == Z.shape and torch.equal(X, Z):
        return var_mean, torch.diagonal(S, dim1=-2, dim2=-1), kl
    Kzx = gm.kernel_matrix(kind, Z, X, ell, outputscale, nu)
    B = Kinv @ Kzx
    os_ = torch.ones(ell.shape[0], dtype=X.dtype) if outputscale is None else outputscale
    mean_f = (B.transpose(-1, -2) @ var_mean.unsqueeze(-1)).squeeze(-1)
    var_f = os_[:, None] - (Kzx * B).sum(-2) + ((Ls.transpose(-1, -2) @ B) ** 2).sum(-2)
    return mean_f, var_f, kl


def variational_elbo(kind, X, Y, Z, ell, var_mean, chol_var, H, task_noise_diag, task_means=None, nu=2.5,
                     outputscale=None, jitter=1e-6, num_data=None, whitened=True):
    """VariationalELBO value (scalar).  H: (q,p); task_noise_diag: (p,) = diag of the task-noise covariance
    incl. global noise; task_means: (p,) constants or None."""
    n, p = Y.shape
    num_data = n if num_data is None else num_data
    pred = latent_predictive if whitened else unwhitened_latent_predictive
    mean_f, var_f, kl = pred(kind, X, Z, ell, var_mean, chol_var, nu, outputscale, jitter)
    mu = mean_f.T @ H                                                        # (n,p)
    if task_means is not None:
        mu = mu + task_means.reshape(1, p)
    var = var_f.T @ (H * H)                                                  # (n,p)
    s = task_noise_diag.reshape(1, p)
    ell_term = -0.5 * (((Y - mu) ** 2 + var) / s + torch.log(s) + math.log(2 * math.pi)).sum()
    return ell_term / n - kl.sum() / num_data
