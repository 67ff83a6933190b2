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


def projected_step(P, X, Y, opt=None, lr=1e-2):
    """ONE full training step of the projected model on the CPU -- the loop body of experiments.py:264-273: zero_grad ->
    model(X) -> -mll -> backward -> AdamW step -- for bench.py's `cpu_baseline` leg ("port").
    The loss is oracle.projected.projected_mll (projected_lmc.py:1178-1241); the q latent exact-GP terms (:1200-1201) use
    `latent_step` above (dense Cholesky + cholesky_inverse, analytic gradient: the cheapest exact CPU form -- torch autograd
    through torch.linalg.cholesky is ~3 x slower) and their gradients are pushed into the autograd graph of the
    projection (:1014-1021) and of the constraint transforms; the projection terms (:1205-1240) go through autograd.
    P: oracle parameter dict whose tensors are leaves; returns (loss, optimiser)."""
    from . import projected as pj
    keys = pj.tensor_keys(P)
    for k in keys:
        P[k].requires_grad_(True)
    if opt is None:
        opt = torch.optim.AdamW([P[k] for k in keys], lr=lr)
    opt.zero_grad()
    n = X.shape[0]
    q = P["n_latents"]
    kind = P["kind"]
    ytil = pj.project_data(P, Y)                                          # q x n, in the graph
    ell, noise, osc = pj.lengthscale(P), pj.projected_noise(P), pj.outputscale(P)
    assert osc is None, "latent_step returns no output-scale gradient: the baseline workload has outputscales=False"
    lps, g_ell, g_nz, g_y = [], [], [], []
    with torch.no_grad():
        for i in range(q):
            lp, ge, gn, gy = latent_step(kind, X, ell[i], noise[i], ytil[i], nu=P["nu"], outputscale=None if osc is None else osc[i])
            lps.append(lp); g_ell.append(ge); g_nz.append(gn); g_y.append(gy)
    # d(-sum lp / n): the analytic latent gradients enter the graph at (lengthscale, noise, projected targets)
    torch.autograd.backward([ell, noise, ytil], [-torch.stack(g_ell) / n, -torch.stack(g_nz).reshape(noise.shape) / n, -torch.stack(g_y) / n])
    terms, const = pj.projection_terms(P, Y)
    proj = sum(terms) + const
    (-proj).backward()
    opt.step()
    loss = -(torch.stack(lps).sum() / n + proj.detach())
    return float(loss), opt
