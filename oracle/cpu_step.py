"""Oracle: memory-bounded CPU restatement of ONE latent's exact MLL + analytic gradient at full
size (n = 8192), used by bench.py's `cpu_baseline` leg ("port") and by full-size spot checks.
TEST INFRASTRUCTURE ONLY -- never imported by the product.

Same arithmetic as gp_math.exact_latent_log_prob_analytic (projected_lmc.py:1200-1201 +
experiments.py:270), but built from row chunks so the n x n x d difference tensor is never
materialised; dense Cholesky / cholesky_inverse from torch (MKL/LAPACK, all host threads).
"""
import math

import torch

from . import gp_math as gm


def latent_step(kind, X, ell, noise, y, nu=2.5, outputscale=None, chunk=512):
    """X (n,d), ell (d,), noise scalar tensor, y (n,) -> (logp, g_ell (d,), g_noise, g_y (n,))."""
    n, d = X.shape
    s = 1.0 if outputscale is None else outputscale
    K = gm.kernel_matrix_chunked(kind, X, X, ell, None, nu, chunk=chunk) * s
    K.diagonal().add_(noise)
    L = torch.linalg.cholesky(K)
    del K
    Kinv = torch.cholesky_inverse(L)
    logdet = 2.0 * torch.log(torch.diagonal(L)).sum()
    del L
    alpha = Kinv @ y
    logp = -0.5 * (alpha @ y + logdet + n * gm.LOG2PI)
    g_noise = 0.5 * (alpha @ alpha - torch.diagonal(Kinv).sum())
    g_ell = torch.zeros(d, dtype=X.dtype)
    U = X / ell
    for s0 in range(0, n, chunk):
        a = U[s0:s0 + chunk]
        diff = a[:, None, :] - U[None, :, :]                       # (c,n,d)
        d2 = diff * diff
        r2 = d2.sum(-1)
        if kind == "rbf":
            base = torch.exp(-0.5 * r2)
        else:
            r = torch.sqrt(r2)
            e = torch.exp(-math.sqrt(2.0 * nu) * r)
            base = (5.0 / 3.0) * (1.0 + math.sqrt(5.0) * r) * e if nu == 2.5 else (3.0 * e if nu == 1.5 else e / r.clamp_min(1e-15))
        Wc = alpha[s0:s0 + chunk, None] * alpha[None, :] - Kinv[s0:s0 + chunk]
        g_ell += torch.einsum("ab,abk->k", Wc * base * s, d2)
    g_ell = 0.5 * g_ell / ell
    return logp, g_ell, g_noise, -alpha


def latent_logp(kind, X, ell, noise, y, nu=2.5, outputscale=None, chunk=512):
    """Forward-only log N(y; 0, K + noise I) at full size (fp64 check value)."""
    n = X.shape[0]
    K = gm.kernel_matrix_chunked(kind, X, X, ell, None, nu, chunk=chunk)
    if outputscale is not None:
        K = K * outputscale
    K.diagonal().add_(noise)
    L = torch.linalg.cholesky(K)
    del K
    z = torch.linalg.solve_triangular(L, y[:, None], upper=False)[:, 0]
    return -0.5 * (z @ z + 2.0 * torch.log(torch.diagonal(L)).sum() + n * gm.LOG2PI)
