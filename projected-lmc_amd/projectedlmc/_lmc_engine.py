"""Host side of the exact dense LMC / ICM path: one (n p) x (n p) factor
buffer, Kronecker-sum assembly kernel, the shared blocked sweep, and the fused gradient kernel.

Replaces gpytorch's LCMKernel / MultitaskKernel evaluation + MultitaskGaussianLikelihood +
MultitaskMultivariateNormal.log_prob + autograd that `MultitaskGPModel` triggers
(projected_lmc.py:462-466, 586-589; experiments.py:184,233,270)."""
import warnings

import torch

from . import _hip, settings
from ._engine import Workspace, LOG2PI, _contig

_ws = {}


def _workspace(N, naug, dtype, device, with_inverse):
    key = (N, naug, dtype, device.index, bool(with_inverse))
    ws = _ws.get(key)
    if ws is None:
        _ws.clear()
        ws = Workspace(N, 1, naug, dtype, device, with_inverse)
        _ws[key] = ws
    return ws


def _factorize(kind, X, ell, osc, B, Sigma, rhs, ws, p, Xs=None):
    """assemble K_full (+ rhs, + cross columns) and factorise; walks gpytorch's jitter ladder on
    failure (psd_safe_cholesky: jitter on the diagonal == on diag(Sigma))."""
    L = _hip.lib()
    dt, dev = ws.dtype, ws.device
    st = _hip.stream_ptr(dev)
    n, d = X.shape
    q = ell.shape[0]
    k = _hip.KIND[kind]
    eye = torch.eye(p, dtype=dt, device=dev)

    def run(jit):
        S = Sigma if jit == 0.0 else (Sigma + jit * eye).contiguous()
        L.call("plmc_lmc_assemble", dt, k, _hip.ptr(X), n, d, p, q, _hip.ptr(ell), _hip.ptr(osc), _hip.ptr(B),
               _hip.ptr(S), _hip.ptr(ws.A), ws.lda, st)
        nrhs = 0 if rhs is None else rhs.shape[1]
        if ws.naug_pad > 0:
            L.call("plmc_write_rhs", dt, _hip.ptr(rhs), nrhs, ws.n, _hip.ptr(ws.A), ws.lda, ws.strideA, 0,
                   ws.naug_pad, 1, st)
        if Xs is not None:
            L.call("plmc_lmc_cross", dt, k, _hip.ptr(X), n, _hip.ptr(Xs), Xs.shape[0], d, p, q, _hip.ptr(ell),
                   _hip.ptr(osc), _hip.ptr(B), _hip.ptr(ws.A), ws.lda, ws.n_pad + nrhs, ws.n_pad, st)
        L.call("plmc_potrf", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.naug, ws.strideA, _hip.ptr(ws.Vd),
               _hip.ptr(ws.logdet), _hip.ptr(ws.info), int(ws.with_inverse), 1, st)

    run(0.0)
    if not settings.check_cholesky.on() or not bool(ws.info.cpu().any()):
        return
    base = settings.cholesky_jitter.value(dt)
    for i in range(settings.cholesky_max_tries.value()):
        jit = base * (10 ** i)
        warnings.warn("A not p.d., added jitter of %.1e to the diagonal" % jit, RuntimeWarning)
        run(jit)
        if not bool(ws.info.cpu().any()):
            return
    raise RuntimeError("Matrix not positive definite after repeatedly adding jitter up to %.1e" % jit)


class LmcExactLogProb(torch.autograd.Function):
    """log N(y; 0, sum_i os_i K_i (x) B_i + I (x) Sigma) with the analytic gradient.

    forward(X (n,d), ell (q,d), oscale (q)|None, B (q,p,p), Sigma (p,p), y (n*p,), kind) -> scalar"""

    @staticmethod
    def forward(ctx, X, ell, oscale, B, Sigma, y, kind):
        _hip.require_device(X, ell, B, Sigma, y)
        L = _hip.lib()
        dt, dev = y.dtype, y.device
        n, d = X.shape
        q, p = B.shape[0], B.shape[-1]
        N = n * p
        need_grad = any(ctx.needs_input_grad[1:6])
        Xc, ellc, osc, Bc, Sc, yc = (_contig(t, dt) for t in (X, ell, oscale, B, Sigma, y))
        ws = _workspace(N, 1, dt, dev, need_grad)
        st = _hip.stream_ptr(dev)
        _factorize(kind, Xc, ellc, osc, Bc, Sc, yc.reshape(1, 1, N), ws, p)
        L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z),
               _hip.ptr(ws.quad), 1, st)
        logp = -0.5 * (ws.quad[0] + ws.logdet[0] + N * LOG2PI)
        if need_grad:
            L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z),
                   _hip.ptr(ws.alpha), 1, st)
            glen = int(L.cdll.plmc_lmc_grad_len(p, q, d))
            grad = torch.empty(glen, dtype=torch.float64, device=dev)
            nbytes = int(L.cdll.plmc_lmc_grad_scratch_bytes(ws.n_pad, p, q, d))
            part = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
            L.call("plmc_lmc_kinv_grad", dt, _hip.KIND[kind], _hip.ptr(ws.W), ws.n_pad, ws.ldw, _hip.ptr(ws.alpha),
                   _hip.ptr(Xc), n, d, p, q, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(Bc), _hip.ptr(grad),
                   _hip.ptr(part), st)
            ctx.save_for_backward(grad, ws.alpha[0, :N].clone())
        ctx.dims = (q, p, d)
        ctx.has_os = oscale is not None
        return logp.to(dt)

    @staticmethod
    def backward(ctx, gout):
        grad, alpha = ctx.saved_tensors
        q, p, d = ctx.dims
        g = gout.to(torch.float64)
        o = 0
        gB = grad[o:o + q * p * p].reshape(q, p, p); o += q * p * p
        gL = grad[o:o + q * d].reshape(q, d); o += q * d
        gO = grad[o:o + q]; o += q
        gS = grad[o:o + p * p].reshape(p, p)
        gB = 0.5 * (gB + gB.transpose(-1, -2))
        gS = 0.5 * (gS + gS.T)
        return (None, g * gL, (g * gO) if ctx.has_os else None, g * gB, g * gS, -(gout.to(alpha.dtype) * alpha), None)


def lmc_exact_log_prob(kind, X, ell, oscale, B, Sigma, y):
    return LmcExactLogProb.apply(X, ell, oscale, B, Sigma, y, kind)


def lmc_posterior(kind, X, ell, oscale, B, Sigma, y, Xs):
    """Posterior task mean (ns,p) and marginal variance of f (ns,p) from one augmented
    factorisation [K_full | y | K_full(X,X*)] (gpytorch prediction strategy behind
    MultitaskGPModel.__call__ in eval mode)."""
    _hip.require_device(X, ell, B, Sigma, y, Xs)
    dt, dev = y.dtype, y.device
    n, d = X.shape
    q, p = B.shape[0], B.shape[-1]
    ns = Xs.shape[0]
    N = n * p
    Xc, Xsc, ellc, osc, Bc, Sc, yc = (_contig(t, dt) for t in (X, Xs, ell, oscale, B, Sigma, y))
    ws = _workspace(N, 1 + ns * p, dt, dev, False)
    _factorize(kind, Xc, ellc, osc, Bc, Sc, yc.reshape(1, 1, N), ws, p, Xs=Xsc)
    aug = ws.A[0, :, ws.n_pad:ws.n_pad + 1 + ns * p]
    z, V = aug[:, 0], aug[:, 1:]
    mean = (V.T @ z).reshape(ns, p)
    os_ = torch.ones(q, dtype=dt, device=dev) if osc is None else osc
    prior_var = (os_[:, None] * torch.diagonal(Bc, dim1=-2, dim2=-1)).sum(0)            # k(x,x) = 1
    var = prior_var[None, :] - (V * V).sum(0).reshape(ns, p)
    return mean, var


def lmc_loo(kind, X, ell, oscale, B, Sigma, y):
    """Leave-one-out moments of the dense LMC / ICM system (MultitaskGPModel.compute_loo, projected_lmc.py:642-656):
    sigma2_i = 1 / [K_full^-1]_ii,  (y - mu_loo)_i = [K_full^-1 y]_i sigma2_i,  both flat (n p,), from the factorisation
    the MLL uses: W = U^-T gives alpha = W^T (U^-T y) and diag(K^-1) = column sums of squares of W (plmc_w_diag)."""
    _hip.require_device(X, ell, B, Sigma, y)
    L = _hip.lib()
    dt, dev = y.dtype, y.device
    n, d = X.shape
    p = B.shape[-1]
    N = n * p
    Xc, ellc, osc, Bc, Sc, yc = (_contig(t, dt) for t in (X, ell, oscale, B, Sigma, y))
    ws = _workspace(N, 1, dt, dev, True)
    _factorize(kind, Xc, ellc, osc, Bc, Sc, yc.reshape(1, 1, N), ws, p)
    st = _hip.stream_ptr(dev)
    L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z), _hip.ptr(ws.quad), 1, st)
    L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z), _hip.ptr(ws.alpha), 1, st)
    kd = torch.empty(1, ws.n_pad, dtype=dt, device=dev)
    L.call("plmc_w_diag", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(kd), 1, st)
    sigma2 = 1.0 / kd[0, :N]
    return sigma2, ws.alpha[0, :N] * sigma2
