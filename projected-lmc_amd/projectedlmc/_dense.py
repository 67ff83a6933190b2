"""Dense SPD systems of moderate size on the blocked sweep of the library (plmc_potrf): the m x m Woodbury factor of
the SGPR / inducing-point latents (sgpr.py; reference: gpytorch's InducingPointKernel behind projected_lmc.py:302-303,
used by every real-data run, realdata_experiments.py:398,505).  Round 1 used torch.linalg.cholesky (rocSOLVER) +
solve_triangular and torch's Cholesky backward here; this is the same factorisation the exact path runs, with the
closed-form adjoint of (quadratic form, log-determinant):
    quad = r^T M^-1 r,  logdet = log det M   ->   dM = -g_q beta beta^T + g_l M^-1,  dr = 2 g_q beta,  beta = M^-1 r,
where M^-1 = W^T W comes from the inverse factor the sweep produces (W = U^-T)."""
import torch

from . import _hip
from ._engine import get_workspace


def _load(ws, M, R):
    """M (q,m,m) -> upper-left block of the factor buffers (identity on the padding), R (q,k,m) -> augmented columns."""
    L = _hip.lib()
    q, m = M.shape[0], M.shape[-1]
    sq = ws.A[:, :, :ws.n_pad]
    sq.zero_()
    sq[:, :m, :m] = M
    if ws.n_pad > m:
        idx = torch.arange(m, ws.n_pad, device=M.device)
        sq[:, idx, idx] = 1.0
    Rc = R.contiguous()                  # (a transposed view makes a copy here: it must outlive the launch that reads it)
    L.call("plmc_write_rhs", ws.dtype, _hip.ptr(Rc), R.shape[1], m, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad,
           q, _hip.stream_ptr(ws.device))
    return Rc


def _factor(ws, what):
    L = _hip.lib()
    L.call("plmc_potrf", ws.dtype, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.naug, ws.strideA, _hip.ptr(ws.Vd), _hip.ptr(ws.logdet),
           _hip.ptr(ws.info), int(ws.with_inverse), ws.q, _hip.stream_ptr(ws.device))
    from . import settings
    if not settings.check_cholesky.on():       # no pivot check wanted: no host sync either
        return
    info = ws.info.cpu()
    if bool(info.any()):
        raise RuntimeError("%s: matrix not positive definite (first failing pivot per matrix: %s)" % (what, info.tolist()))


class SpdQuadLogdet(torch.autograd.Function):
    """forward(M (q,m,m) SPD, r (q,m)) -> (quad (q,), logdet (q,))."""

    @staticmethod
    def forward(ctx, M, r):
        _hip.require_device(M, r)
        L = _hip.lib()
        dt, dev = M.dtype, M.device
        q, m = M.shape[0], M.shape[-1]
        need = any(ctx.needs_input_grad)
        ws = get_workspace(m, q, 1, dt, dev, True)
        keep = _load(ws, M.detach(), r.detach().reshape(q, 1, m))      # noqa: F841 (the copy lives until the sweep is queued)
        _factor(ws, "SpdQuadLogdet")
        st = _hip.stream_ptr(dev)
        L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z), _hip.ptr(ws.quad), q, st)
        quad, logdet = ws.quad.to(dt, copy=True), ws.logdet.to(dt, copy=True)      # the workspace is reused
        if need:
            L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z), _hip.ptr(ws.alpha), q, st)
            W = torch.tril(ws.W[:, :m, :m])
            ctx.save_for_backward(ws.alpha[:, :m].clone(), W.transpose(-1, -2) @ W)          # beta, M^-1
        return quad, logdet

    @staticmethod
    def backward(ctx, gq, gl):
        beta, Minv = ctx.saved_tensors
        gM = gl[:, None, None] * Minv - gq[:, None, None] * (beta.unsqueeze(-1) * beta.unsqueeze(-2))
        return gM, 2.0 * gq[:, None] * beta


def spd_quad_logdet(M, r):
    return SpdQuadLogdet.apply(M, r)


def spd_half_solve(M, R):
    """U^-T R for M = U^T U (q,m,m), R (q,m,k): the forward substitution L^-1 R of a Cholesky factor L = U^T, read off
    the augmented columns of one sweep.  No gradient (prediction path)."""
    _hip.require_device(M, R)
    dt, dev = M.dtype, M.device
    q, m, k = R.shape
    ws = get_workspace(m, q, k, dt, dev, False)
    with torch.no_grad():
        keep = _load(ws, M.detach(), R.detach().transpose(-1, -2))      # noqa: F841 (the copy lives until the sweep is queued)
        _factor(ws, "spd_half_solve")
        return ws.A[:, :m, ws.n_pad:ws.n_pad + k].clone()


def _padded(t, rows, cols):
    q, r, c = t.shape
    if r == rows and c == cols and t.is_contiguous():
        return t
    out = torch.zeros(q, rows, cols, dtype=t.dtype, device=t.device)
    out[:, :r, :c] = t
    return out


TRI_A_LOWER, TRI_B_LOWER, TRI_A_UPPER, TRI_C_LOWER, TRI_C_ZERO = 1, 2, 4, 8, 16      # include/plmc.h PLMC_TRI_*


def gemm_tn(A_km, B_kn, tri=0):
    """C = A^T B for batches of K-major operands A (q,K,M), B (q,K,N) -> (q,M,N) on the library's tile engine
    (plmc_gemm_tn): operands are zero-padded to block multiples (a transposed view is made contiguous by the same copy).
    tri: PLMC_TRI_* bits declaring triangular operands / a lower-triangular result (plmc_gemm_tn_tri: only the contraction
    range with entries is walked; with TRI_C_LOWER alone the tiles above the block diagonal hold garbage -- take torch.tril).
    No autograd: used inside hand-written backward passes (`_var_engine`)."""
    _hip.require_device(A_km, B_kn)
    L = _hip.lib()
    dt, dev = A_km.dtype, A_km.device
    q, K, M = A_km.shape
    N = B_kn.shape[-1]
    nb = L.cdll.plmc_block()
    Kp, Mp, Np = -(-K // 16) * 16, -(-M // nb) * nb, -(-N // nb) * nb
    A = _padded(A_km.detach(), Kp, Mp)
    B = _padded(B_kn.detach().to(dt), Kp, Np)
    C = torch.empty(q, Mp, Np, dtype=dt, device=dev)
    if tri:
        L.call("plmc_gemm_tn_tri", dt, 0, int(tri), Mp, Np, Kp, _hip.ptr(A), Mp, Kp * Mp, _hip.ptr(B), Np, Kp * Np, _hip.ptr(C), Np, Mp * Np, q,
               _hip.stream_ptr(dev))
    else:
        L.call("plmc_gemm_tn", dt, 0, Mp, Np, Kp, _hip.ptr(A), Mp, Kp * Mp, _hip.ptr(B), Np, Kp * Np, _hip.ptr(C), Np, Mp * Np, q,
               _hip.stream_ptr(dev))
    return C[:, :M, :N]
