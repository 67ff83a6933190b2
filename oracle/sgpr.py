"""Oracle: sparse GP regression (Titsias) MLL and predictive moments.  TEST INFRASTRUCTURE ONLY.

Restates what `InducingPointKernel` (projected_lmc.py:302-303) makes gpytorch compute
[gpytorch-knowledge, unverified offline]:
    Q = K_xz K_zz^-1 K_zx,
    mll * n = log N(y; 0, Q + s I)  -  1/2 sum_i (k_ii - q_ii) / s      (added loss term)
    predictive mean = Q_*x (Q + s I)^-1 y,   covariance = Q_** - Q_*x (Q + s I)^-1 Q_x*.
Dense, direct formulas (no Woodbury) so that it is independent of the product's algebra.
"""
import torch

from . import gp_math as gm


def sgpr_terms(kind, X, Z, ell, noise, y, nu=2.5, outputscale=None):
    """Per latent: (log N(y; 0, Q + sI), trace term).  ell (q,d), noise (q,), y (q,n)."""
    Kzz = gm.kernel_matrix(kind, Z, Z, ell, outputscale, nu)
    Kzx = gm.kernel_matrix(kind, Z, X, ell, outputscale, nu)
    Q = Kzx.transpose(-1, -2) @ torch.linalg.solve(Kzz, Kzx)
    n = X.shape[0]
    C = Q + noise.reshape(-1, 1, 1) * torch.eye(n, dtype=X.dtype)
    lp = gm.mvn_log_prob(C, y)
    os_ = torch.ones(ell.shape[0], dtype=X.dtype) if outputscale is None else outputscale
    trace = -0.5 * (n * os_ - torch.diagonal(Q, dim1=-2, dim2=-1).sum(-1)) / noise
    return lp, trace


def sgpr_posterior(kind, X, Z, ell, noise, y, Xs, nu=2.5, outputscale=None):
    Kzz = gm.kernel_matrix(kind, Z, Z, ell, outputscale, nu)
    Kzx = gm.kernel_matrix(kind, Z, X, ell, outputscale, nu)
    Kzs = gm.kernel_matrix(kind, Z, Xs, ell, outputscale, nu)
    n = X.shape[0]
    Q = Kzx.transpose(-1, -2) @ torch.linalg.solve(Kzz, Kzx)
    Qsx = Kzs.transpose(-1, -2) @ torch.linalg.solve(Kzz, Kzx)
    Qss = Kzs.transpose(-1, -2) @ torch.linalg.solve(Kzz, Kzs)
    C = Q + noise.reshape(-1, 1, 1) * torch.eye(n, dtype=X.dtype)
    mean = (Qsx @ torch.linalg.solve(C, y.unsqueeze(-1))).squeeze(-1)
    cov = Qss - Qsx @ torch.linalg.solve(C, Qsx.transpose(-1, -2))
    return mean, cov
