"""Host side of the variational (SVGP-LMC) path.

The heavy arithmetic of gpytorch's whitened VariationalStrategy [gpytorch-knowledge] -- K_ZZ, its
Cholesky factor, K_ZX and the interpolation term A = L^-1 K_ZX for q latent GPs -- is ONE augmented
factorisation [K_ZZ + jit I | K_ZX] of the HIP sweep (csrc/potrf.hip): assembly, factorisation and
the forward solve all happen in the library, the result A is read from the augmented block.

Backward (what `loss.backward()`, experiments.py:270, derives through gpytorch): with
G = d loss / d A,   Cbar = L^-T G,   Lbar = -tril(Cbar A^T),   Kbar = sym(L^-T Phi(L^T Lbar) L^-1)
(standard Cholesky adjoint; L^-1 = W comes from the same sweep, `with_inverse`), then the
kernel-matrix adjoints Kbar, Cbar are pulled back to lengthscales / outputscales / inducing
locations by the fused HIP kernel `plmc_kernel_vjp`.  The rectangular m x m x n products of the
adjoint are plain library GEMMs (torch.matmul -> rocBLAS/hipBLASLt).
"""
import torch

from . import _hip
from ._engine import Workspace, _contig

_ws = {}


def _workspace(m, q, n, dtype, device, with_inverse):
    key = (m, q, n, dtype, device.index, bool(with_inverse))
    ws = _ws.get(key)
    if ws is None:
        _ws.clear()
        ws = Workspace(m, q, n, dtype, device, with_inverse)
        _ws[key] = ws
    return ws


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
        L.call("plmc_potrf", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, n, ws.strideA, _hip.ptr(ws.Vd),
               _hip.ptr(ws.logdet), _hip.ptr(ws.info), int(need_grad), q, st)
        info = ws.info.cpu()
        if bool(info.any()):
            raise RuntimeError("K_ZZ + jitter not positive definite (first failing pivot per latent: %s)" % info.tolist())
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
        A, U, W, Z, X, ell, osc = ctx.saved_tensors
        osc = osc if ctx.has_os else None
        kind = ctx.kind
        G = G.contiguous()
        Cbar = W.transpose(-1, -2) @ G                                          # L^-T G          (q,m,n)
        Lbar = -torch.tril(Cbar @ A.transpose(-1, -2))                          # (q,m,m)
        P = torch.tril(U @ Lbar)                                                # Phi(L^T Lbar)
        P = P - 0.5 * torch.diag_embed(torch.diagonal(P, dim1=-2, dim2=-1))
        Kbar = W.transpose(-1, -2) @ P @ W
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
