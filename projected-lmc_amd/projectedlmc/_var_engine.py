"""Host side of the variational (SVGP-LMC) path.

The heavy arithmetic of gpytorch's whitened VariationalStrategy [gpytorch-knowledge] -- K_ZZ, its
Cholesky factor, K_ZX and the interpolation term A = L^-1 K_ZX for q latent GPs -- is ONE augmented
factorisation [K_ZZ + jit I | K_ZX] of the HIP sweep (csrc/potrf.hip): assembly, factorisation and
the forward solve all happen in the library, the result A is read from the augmented block.

Backward (what `loss.backward()`, experiments.py:270, derives through gpytorch): with
G = d loss / d A,   Cbar = L^-T G,   Lbar = -tril(Cbar A^T),   Kbar = sym(L^-T Phi(L^T Lbar) L^-1)
(standard Cholesky adjoint; L^-1 = W comes from the same sweep, `with_inverse`), then the
kernel-matrix adjoints Kbar, Cbar are pulled back to lengthscales / outputscales / inducing
locations by the fused HIP kernel `plmc_kernel_vjp`.  The m x m x n products of the adjoint run on the library's own
tile engine (`plmc_gemm_tn` through `_dense.gemm_tn`; every product is written in "TN" form, C = X^T Y).
"""
import torch

from . import _hip
from ._engine import Workspace, _contig, _DeferredInfo, _check_chain_abort

_ws = {}


def _workspace(m, q, n, dtype, device, with_inverse):
    key = (m, q, n, dtype, device.index, bool(with_inverse))
    ws = _ws.get(key)
    if ws is None:
        _ws.clear()
        ws = Workspace(m, q, n, dtype, device, with_inverse)
        _ws[key] = ws
    return ws


def _raise_if_not_pd(check):
    """Wait for the deferred pivot check of a sweep (a copy of `info` in pinned memory + its event) and raise as the eager check did."""
    if check.failed():
        raise RuntimeError("K_ZZ + jitter not positive definite (first failing pivot per latent: %s)" % check.host.tolist())


def kernel_vjp(kind, X1, X2, ell, oscale, G):
    """(gX1 (n1,d) summed over latents, gEll (q,d), gOs (q)) in fp64 for K_i = os_i k(X1, X2; ell_i)."""
    L = _hip.lib()
    dt, dev = G.dtype, G.device
    q, n1, n2 = G.shape
    d = X1.shape[1]
    G = G.contiguous()
    gX = torch.empty(q, n1, d, dtype=torch.float64, device=dev)
    gE = torch.empty(q, n1, d, dtype=torch.float64, device=dev)
    gO = torch.empty(q, n1, dtype=torch.float64, device=dev)
    L.call("plmc_kernel_vjp", dt, _hip.KIND[kind], _hip.ptr(X1), n1, _hip.ptr(X2), n2, d, _hip.ptr(ell),
           _hip.ptr(oscale), _hip.ptr(G), n2, n1 * n2, _hip.ptr(gX), _hip.ptr(gE), _hip.ptr(gO), q,
           _hip.stream_ptr(dev))
    return gX.sum(0), gE.sum(1), gO.sum(1)


class WhitenedInterp(torch.autograd.Function):
    """A_i = L_i^-1 K_i(Z, X),  L_i L_i^T = K_i(Z, Z) + jitter I   for q latent kernels.

    forward(Z (m,d), X (n,d), ell (q,d), oscale (q)|None, kind, jitter) -> A (q, m, n)"""

    @staticmethod
    def forward(ctx, Z, X, ell, oscale, kind, jitter):
        _hip.require_device(Z, X, ell)
        L = _hip.lib()
        dt, dev = ell.dtype, ell.device
        m, d = Z.shape
        n = X.shape[0]
        q = ell.shape[0]
        need_grad = any(ctx.needs_input_grad[:4])
        Zc, Xc, ellc, osc = (_contig(t, dt) for t in (Z, X, ell, oscale))
        ws = _workspace(m, q, n, dt, dev, need_grad)
        st = _hip.stream_ptr(dev)
        k = _hip.KIND[kind]
        jit = torch.full((q,), float(jitter), dtype=dt, device=dev)
        L.call("plmc_assemble", dt, k, _hip.ptr(Zc), m, d, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(jit),
               _hip.ptr(ws.A), ws.lda, ws.strideA, q, st)
        L.call("plmc_write_rhs", dt, None, 0, m, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad, q, st)
        L.call("plmc_assemble_cross", dt, k, _hip.ptr(Zc), m, _hip.ptr(Xc), n, d, _hip.ptr(ellc), _hip.ptr(osc),
               _hip.ptr(ws.A), ws.lda, ws.strideA, ws.n_pad, ws.n_pad, q, st)
        # eig_lo = the jitter: lambda_min(K_ZZ + jitter I) >= jitter (bound for the fp16 split of the bulk fp32 products)
        L.call("plmc_potrf_ex", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, n, ws.strideA, _hip.ptr(ws.Vd),
               _hip.ptr(ws.logdet), _hip.ptr(ws.info), int(need_grad), q, _hip.ptr(jit), st)
        # the pivot check does not stall the step: `info` goes to pinned host memory behind the sweep and is looked at in
        # backward() (the whole forward pass -- the ELBO's contractions -- is queued by then; the pattern of
        # _engine._DeferredInfo); without a backward pass (no gradient needed) it is looked at now
        check = _DeferredInfo(ws)
        ctx.pivot_check = check if need_grad else None
        if not need_grad:
            _raise_if_not_pd(check)
        A = ws.A[:, :m, ws.n_pad:ws.n_pad + n].clone()
        if need_grad:
            U = torch.triu(ws.A[:, :m, :m])                                     # L^T
            W = torch.tril(ws.W[:, :m, :m])                                     # L^-1
            ctx.save_for_backward(A, U, W, Zc, Xc, ellc, osc if osc is not None else torch.empty(0, device=dev))
        ctx.kind = kind
        ctx.has_os = oscale is not None
        return A

    @staticmethod
    def backward(ctx, G):
        if ctx.pivot_check is not None:
            _raise_if_not_pd(ctx.pivot_check)
        A, U, W, Z, X, ell, osc = ctx.saved_tensors
        osc = osc if ctx.has_os else None
        kind = ctx.kind
        G = G.contiguous()
        from ._dense import gemm_tn, TRI_A_LOWER as AL, TRI_B_LOWER as BL, TRI_A_UPPER as AU, TRI_C_LOWER as CL, TRI_C_ZERO as CZ
        mT = lambda t: t.transpose(-1, -2)
        # every factor here is triangular (W = L^-1 and the adjoints lower, U = L^T upper): the products only walk the contraction
        # range with entries (plmc_gemm_tn_tri) -- 768 -> ~280 GFLOP at BASELINE config 4 (m = 2000, n = 3000, q = 8)
        Cbar = gemm_tn(W, G, AL)                                                # L^-T G = W^T G  (q,m,n)
        Lbar = -torch.tril(gemm_tn(mT(Cbar), mT(A), CL))                        # -tril(Cbar A^T) (q,m,m)
        P = torch.tril(gemm_tn(mT(U), Lbar, AL | BL | CL))                      # Phi(L^T Lbar),  L^T = U
        P = P - 0.5 * torch.diag_embed(torch.diagonal(P, dim1=-2, dim2=-1))
        Kbar = gemm_tn(W, gemm_tn(mT(P), W, AU | BL | CL | CZ), AL | BL)        # W^T (P W)
        Kbar = 0.5 * (Kbar + Kbar.transpose(-1, -2))
        # pull the kernel-matrix adjoints back to (Z, ell, oscale); Z enters K_ZZ through both arguments
        gZ1, gE1, gO1 = kernel_vjp(kind, Z, Z, ell, osc, Kbar)
        gZ2, gE2, gO2 = kernel_vjp(kind, Z, X, ell, osc, Cbar)
        gZ = 2.0 * gZ1 + gZ2                                                    # Kbar symmetric: both roles equal
        gE = gE1 + gE2
        gO = (gO1 + gO2) if ctx.has_os else None
        dt = A.dtype
        return gZ.to(dt), None, gE.to(dt), None if gO is None else gO.to(dt), None, None


def whitened_interp(kind, Z, X, ell, oscale, jitter):
    return WhitenedInterp.apply(Z, X, ell, oscale, kind, jitter)


class LowerTMatmul(torch.autograd.Function):
    """Ls^T A for a batch of LOWER-triangular Ls (q,m,m) and A (q,m,n) -- the S-dependent part of the whitened strategy's
    predictive covariance, Bm = Ls^T A (gpytorch VariationalStrategy [gpytorch-knowledge]; reached from projected_lmc.py:672-683) --
    with its adjoints, on the library's tile engine with the triangular contraction ranges (plmc_gemm_tn_tri): half the
    flops of the three dense m x m x n products torch.bmm and its autograd would run.  `Ls` must be lower triangular (the caller
    passes chol_variational_covar.tril()); the gradient returned for it is the lower triangle."""

    @staticmethod
    def forward(ctx, Ls, A):
        from ._dense import gemm_tn, TRI_A_LOWER
        ctx.save_for_backward(Ls, A)
        return gemm_tn(Ls, A, TRI_A_LOWER)

    @staticmethod
    def backward(ctx, G):
        from ._dense import gemm_tn, TRI_A_UPPER, TRI_C_LOWER
        Ls, A = ctx.saved_tensors
        mT = lambda t: t.transpose(-1, -2)
        gA = gemm_tn(mT(Ls), G, TRI_A_UPPER) if ctx.needs_input_grad[1] else None        # Ls G
        gLs = torch.tril(gemm_tn(mT(A), mT(G), TRI_C_LOWER)) if ctx.needs_input_grad[0] else None   # tril(A G^T)
        return gLs, gA


def lower_t_matmul(Ls, A):
    return LowerTMatmul.apply(Ls, A)


# ------------------------------------------------------------------------------------------------
# Unwhitened strategy (train_ind_ratio == 1, projected_lmc.py:724-729): q(u) = N(m, Ls Ls^T) against the
# prior N(0, Khat), Khat = K_ZZ + jitter I, with Z = the training inputs (n x n matrices per latent).
class GaussianKLToKernelPrior(torch.autograd.Function):
    """KL( N(m_i, Ls_i Ls_i^T) || N(0, Khat_i) ) = 1/2 [ tr(Khat^-1 S) + m^T Khat^-1 m - n + log det Khat - log det S ]
    for q latent kernels, with the analytic gradient.

    forward(Z (n,d), ell (q,d), oscale (q)|None, mvar (q,n), Ls (q,n,n) lower, kind, jitter) -> (q,)

    One augmented sweep [Khat | m, Ls] gives U^-T [m, Ls] (trace and quadratic form are its squared norms),
    log det Khat and W = U^-T.  Backward: Khat^-1 [m, Ls] = W^T (U^-T [m, Ls]) and
    dKL/dKhat = 1/2 (Khat^-1 - (Khat^-1 [m, Ls]) (Khat^-1 [m, Ls])^T) are plain library GEMMs; the kernel-matrix
    adjoint is pulled back to (Z, ell, oscale) by `plmc_kernel_vjp`."""

    @staticmethod
    def forward(ctx, Z, ell, oscale, mvar, Ls, kind, jitter):
        _hip.require_device(Z, ell, mvar, Ls)
        L = _hip.lib()
        dt, dev = ell.dtype, ell.device
        n, d = Z.shape
        q = ell.shape[0]
        Zc, ellc, osc = (_contig(t, dt) for t in (Z, ell, oscale))
        rhs = torch.cat([mvar.detach().to(dt).unsqueeze(-1), torch.tril(Ls.detach().to(dt))], -1)     # (q, n, n+1)
        rhs_t = rhs.transpose(-1, -2).contiguous()                                                  # write_rhs wants (q, nrhs, n)
        ws = _workspace(n, q, n + 1, dt, dev, True)
        st = _hip.stream_ptr(dev)
        jit = torch.full((q,), float(jitter), dtype=dt, device=dev)
        L.call("plmc_assemble", dt, _hip.KIND[kind], _hip.ptr(Zc), n, d, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(jit),
               _hip.ptr(ws.A), ws.lda, ws.strideA, q, st)
        L.call("plmc_write_rhs", dt, _hip.ptr(rhs_t), n + 1, n, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad, q, st)
        L.call("plmc_potrf_ex", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, n + 1, ws.strideA, _hip.ptr(ws.Vd),
               _hip.ptr(ws.logdet), _hip.ptr(ws.info), 1, q, _hip.ptr(jit), st)
        check = _DeferredInfo(ws)                                                      # looked at in backward() (see WhitenedInterp)
        ctx.pivot_check = check if any(ctx.needs_input_grad[:5]) else None
        if ctx.pivot_check is None:
            _raise_if_not_pd(check)
        Zs = ws.A[:, :n, ws.n_pad:ws.n_pad + n + 1]                                  # U^-T [m, Ls]
        sq = (Zs.double() ** 2).sum(-2)                                              # (q, n+1) column norms
        quad, tr = sq[:, 0], sq[:, 1:].sum(-1)
        dg = torch.diagonal(Ls.detach(), dim1=-2, dim2=-1).double()
        logdetS = 2.0 * torch.log(dg.abs()).sum(-1)
        kl = 0.5 * (tr + quad - n + ws.logdet - logdetS)
        W = torch.tril(ws.W[:, :n, :n])                                              # U^-T (lower)
        ctx.save_for_backward(W, Zs.clone(), Zc, ellc, osc if osc is not None else torch.empty(0, device=dev), dg)
        ctx.kind, ctx.has_os = kind, oscale is not None
        return kl.to(dt)

    @staticmethod
    def backward(ctx, g):
        if ctx.pivot_check is not None:
            _raise_if_not_pd(ctx.pivot_check)
        W, Zs, Z, ell, osc, dg = ctx.saved_tensors
        osc = osc if ctx.has_os else None
        dt = W.dtype
        from ._dense import gemm_tn, TRI_A_LOWER, TRI_B_LOWER
        Af = gemm_tn(W, Zs, TRI_A_LOWER)                                             # Khat^-1 [m, Ls] = W^T Zs (q, n, n+1)
        g_m = Af[..., 0]
        g_Ls = torch.tril(Af[..., 1:]) - torch.diag_embed((1.0 / dg).to(dt))
        Kinv = gemm_tn(W, W, TRI_A_LOWER | TRI_B_LOWER)
        Gk = 0.5 * (Kinv - gemm_tn(Af.transpose(-1, -2), Af.transpose(-1, -2))) * g.to(dt)[:, None, None]
        gZ, gE, gO = kernel_vjp(ctx.kind, Z, Z, ell, osc, Gk)
        gl = g.to(dt)
        return ((2.0 * gZ).to(dt), gE.to(dt), gO.to(dt) if ctx.has_os else None, gl[:, None] * g_m, gl[:, None, None] * g_Ls,
                None, None)


def gaussian_kl_to_kernel_prior(kind, Z, ell, oscale, mvar, Ls, jitter):
    return GaussianKLToKernelPrior.apply(Z, ell, oscale, mvar, Ls, kind, jitter)


def prior_cholesky(kind, Z, ell, oscale, jitter):
    """Lower Cholesky factors of K_ZZ + jitter I (q, n, n), no gradient: the first-call initialisation of q(u)."""
    L = _hip.lib()
    dt, dev = ell.dtype, ell.device
    n, d = Z.shape
    q = ell.shape[0]
    Zc, ellc, osc = (_contig(t, dt) for t in (Z, ell, oscale))
    ws = _workspace(n, q, 0, dt, dev, False)
    st = _hip.stream_ptr(dev)
    jit = torch.full((q,), float(jitter), dtype=dt, device=dev)
    L.call("plmc_assemble", dt, _hip.KIND[kind], _hip.ptr(Zc), n, d, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(jit),
           _hip.ptr(ws.A), ws.lda, ws.strideA, q, st)
    L.call("plmc_potrf_ex", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, 0, ws.strideA, _hip.ptr(ws.Vd), _hip.ptr(ws.logdet),
           _hip.ptr(ws.info), 0, q, _hip.ptr(jit), st)
    if bool(ws.info.cpu().any()):
        raise RuntimeError("K_ZZ + jitter not positive definite")
    return torch.triu(ws.A[:, :n, :n]).transpose(-1, -2).contiguous()


def unwhitened_predictive(kind, Z, X, ell, oscale, mvar, Ls, jitter):
    """Marginal q(f(X)) for X != Z (eval mode), no gradient:  B = Khat^-1 K_ZX,
    mean = B^T m,  var = k(x,x) - colsum((U^-T K_ZX)^2) + colsum((Ls^T B)^2)."""
    L = _hip.lib()
    dt, dev = ell.dtype, ell.device
    n, d = Z.shape
    ns = X.shape[0]
    q = ell.shape[0]
    Zc, Xc, ellc, osc = (_contig(t, dt) for t in (Z, X, ell, oscale))
    ws = _workspace(n, q, ns, dt, dev, True)
    st = _hip.stream_ptr(dev)
    k = _hip.KIND[kind]
    jit = torch.full((q,), float(jitter), dtype=dt, device=dev)
    L.call("plmc_assemble", dt, k, _hip.ptr(Zc), n, d, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(jit), _hip.ptr(ws.A),
           ws.lda, ws.strideA, q, st)
    L.call("plmc_write_rhs", dt, None, 0, n, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad, q, st)
    L.call("plmc_assemble_cross", dt, k, _hip.ptr(Zc), n, _hip.ptr(Xc), ns, d, _hip.ptr(ellc), _hip.ptr(osc),
           _hip.ptr(ws.A), ws.lda, ws.strideA, ws.n_pad, ws.n_pad, q, st)
    L.call("plmc_potrf_ex", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ns, ws.strideA, _hip.ptr(ws.Vd), _hip.ptr(ws.logdet),
           _hip.ptr(ws.info), 1, q, _hip.ptr(jit), st)
    if bool(ws.info.cpu().any()):
        raise RuntimeError("K_ZZ + jitter not positive definite")
    C = ws.A[:, :n, ws.n_pad:ws.n_pad + ns]                                          # U^-T K_ZX
    from ._dense import gemm_tn, TRI_A_LOWER
    B = gemm_tn(torch.tril(ws.W[:, :n, :n]), C, TRI_A_LOWER)                         # Khat^-1 K_ZX = W^T C
    mean = (B.transpose(-1, -2) @ mvar.to(dt).unsqueeze(-1)).squeeze(-1)
    os_ = torch.ones(q, dtype=dt, device=dev) if osc is None else osc
    LB = gemm_tn(torch.tril(Ls.to(dt)), B, TRI_A_LOWER)                               # Ls^T B
    var = os_[:, None] - (C * C).sum(-2) + (LB * LB).sum(-2)
    return mean, var
