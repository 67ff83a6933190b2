"""Host side of the exact-GP hot path: torch-owned workspaces + autograd wrappers around the
C ABI (include/plmc.h).  PyTorch is plumbing here (device memory, streams, autograd graph);
all n x n work happens in the HIP library.

What it replaces in the reference: the gpytorch evaluation chain behind
`latent_output.log_prob(proj_target)` (projected_lmc.py:1200-1201), ExactMarginalLogLikelihood
(experiments.py:233) and `loss.backward()` (experiments.py:270); see SURVEY.md 8a rows a1-a4.
"""
import math
import os
import warnings

import torch

from . import _hip
from . import settings

LOG2PI = math.log(2.0 * math.pi)


class Workspace:
    """Caller-owned device buffers for q latent GPs on n points with naug augmented columns.
    Layout documented in include/plmc.h."""

    def __init__(self, n, q, naug, dtype, device, with_inverse=True, keep_planes=False):
        L = _hip.lib()
        self.n, self.q, self.naug, self.dtype, self.device = n, q, naug, dtype, device
        self.with_inverse = bool(with_inverse)
        # keep_planes (fp32, with the inverse factor): the sweep keeps the 16-bit planes of every group's solved rows, so that new
        # augmented columns can be forward-substituted on the split engine (plmc_potrs_aug_kept): the eval-mode cache
        self.keep_planes = bool(keep_planes) and self.with_inverse and dtype == torch.float32
        self.NB = L.cdll.plmc_block()
        self.n_pad = int(L.cdll.plmc_pad(n))
        self.naug_pad = int(L.cdll.plmc_pad(naug)) if naug > 0 else 0
        self.wcol0 = self.n_pad + self.naug_pad
        self.lda = self.wcol0 + (self.n_pad if with_inverse else 0)
        if (self.lda // self.NB) % 2 == 0:
            self.lda += self.NB          # odd number of blocks per row: keeps the row stride off a power of two
        self.strideA = self.n_pad * self.lda
        esz = torch.empty((), dtype=dtype).element_size()
        # scratch sizes depend on (n_pad, lda, element size) only -- never on a dev knob (include/plmc.h, version 3)
        vd_blocks = L.cdll.plmc_vd_blocks_keep(self.n_pad, self.lda) if self.keep_planes else L.cdll.plmc_vd_blocks_for(self.n_pad, self.lda, esz)
        self.Vd = torch.empty(q, int(vd_blocks), self.NB, self.NB, dtype=dtype, device=device)
        self.m = self.n_pad // self.NB
        self.A = torch.empty(q, self.n_pad, self.lda, dtype=dtype, device=device)
        self.logdet = torch.empty(q, dtype=torch.float64, device=device)
        self.quad = torch.empty(q, dtype=torch.float64, device=device)
        self.info = torch.empty(q, dtype=torch.int32, device=device)
        self.z = torch.empty(q, self.n_pad, dtype=dtype, device=device)
        self.W = self.alpha = self.partials = None
        if with_inverse:
            # W = U^-T lives in the same buffer, right of the augmented block (include/plmc.h)
            self.W = self.A[:, :, self.wcol0:self.wcol0 + self.n_pad]
            self.ldw, self.strideW = self.lda, self.strideA
            self.alpha = torch.empty(q, self.n_pad, dtype=dtype, device=device)
            # per-tile partial sums of the gradient kernel; the 16-bit planes of W it multiplies are left in Vd by the sweep
            # (plmc_kinv_grad_vd), except with PLMC_SPLIT=0 / fp64, which need none
            nbytes = int(L.cdll.plmc_grad_partials_bytes(self.n_pad, q))
            self.partials = torch.empty(nbytes // 8, dtype=torch.float64, device=device)


_ws_cache = {}


def _release(ws):
    """Before a workspace's memory goes back to the caching allocator: the gradient kernel of its last evaluation may
    still read W / alpha and write the partials on the gradient stream -- make the allocating (current) stream wait."""
    pending = getattr(ws, "pending", None)
    if pending is not None:
        torch.cuda.current_stream(ws.device).wait_event(pending)
        ws.pending = None


def _drop_all():
    for ws in _ws_cache.values():
        _release(ws)
    _ws_cache.clear()


def get_workspace(n, q, naug, dtype, device, need_grad):
    key = (n, q, naug, dtype, device.index, bool(need_grad))
    ws = _ws_cache.get(key)
    if ws is None:
        if len(_ws_cache) > 3:
            _drop_all()
        ws = Workspace(n, q, naug, dtype, device, need_grad)
        _ws_cache[key] = ws
    # a gradient kernel of the previous evaluation may still be reading this workspace on the gradient stream
    pending = getattr(ws, "pending", None)
    if pending is not None:
        torch.cuda.current_stream(device).wait_event(pending)
        ws.pending = None
    return ws


_grad_streams = {}


def grad_stream(device):
    """Stream the fused K^-1 + gradient kernel runs on (one per device; PLMC_GRAD_STREAM=0: the caller's stream).
    The kernel is the last consumer of the factor and nothing in the forward pass needs its output, so the rest of
    the forward pass and the part of the backward pass ahead of the latent node (the projection terms of
    ProjectedLMCmll) -- dozens of launch-bound torch kernels -- run beside it instead of behind it."""
    if os.environ.get("PLMC_GRAD_STREAM", "1") == "0":
        return None
    s = _grad_streams.get(device.index)
    if s is None:
        s = _grad_streams[device.index] = torch.cuda.Stream(device)
    return s


def free_workspaces():
    _drop_all()


def _contig(t, dtype=None):
    if t is None:
        return None
    t = t.detach()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def factorize(kind, X, ell, oscale, noise, rhs, ws, Xs=None, kacc=False):
    """Assemble Khat (+ rhs / cross-covariance columns) and run the blocked Cholesky.
    rhs: (q, nrhs, n) or None.  Returns nothing; results live in ws (A, Vd, logdet, info).
    kacc: also accumulate Khat^-1 = W^T W inside the sweep (plmc_potrf with_inverse = 2; needs ws.with_inverse)."""
    L = _hip.lib()
    dt, dev = ws.dtype, ws.device
    st = _hip.stream_ptr(dev)
    k = _hip.KIND[kind]
    q, n, d = ws.q, ws.n, X.shape[1]
    # round 4: assembly and sweep as ONE library call (plmc_factorize_ex_*): the sweep writes the rows of its first group itself and
    # queues the others beside that group's chain (same kernels, same data: bit-identical; PLMC_FUSED_ASSEMBLE=0: two calls)
    fused = os.environ.get("PLMC_FUSED_ASSEMBLE", "1") != "0"
    if not fused:
        L.call("plmc_assemble", dt, k, _hip.ptr(X), n, d, _hip.ptr(ell), _hip.ptr(oscale), _hip.ptr(noise),
               _hip.ptr(ws.A), ws.lda, ws.strideA, q, st)
    nrhs = 0 if rhs is None else rhs.shape[1]
    if ws.naug_pad > 0:
        L.call("plmc_write_rhs", dt, _hip.ptr(rhs), nrhs, n, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad, q, st)
    if Xs is not None:
        L.call("plmc_assemble_cross", dt, k, _hip.ptr(X), n, _hip.ptr(Xs), Xs.shape[0], d, _hip.ptr(ell),
               _hip.ptr(oscale), _hip.ptr(ws.A), ws.lda, ws.strideA, ws.n_pad + nrhs, ws.n_pad, q, st)
    # eig_lo = the noise variances: lambda_min(K + s2 I) >= s2 -- the bound the two-plane fp16 split of the bulk fp32 products
    # scales its operands with (include/plmc.h, plmc_potrf_ex_*); ignored by the fp64 entry point
    flags = ((2 if kacc else 1) | (4 if ws.keep_planes else 0)) if ws.with_inverse else 0
    if fused:
        L.call("plmc_factorize_ex", dt, k, _hip.ptr(X), n, d, _hip.ptr(ell), _hip.ptr(oscale), _hip.ptr(noise), _hip.ptr(ws.A), ws.n_pad,
               ws.lda, ws.naug, ws.strideA, _hip.ptr(ws.Vd), _hip.ptr(ws.logdet), _hip.ptr(ws.info), flags, q, _hip.ptr(noise), st)
    else:
        L.call("plmc_potrf_ex", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.naug, ws.strideA, _hip.ptr(ws.Vd),
               _hip.ptr(ws.logdet), _hip.ptr(ws.info), flags, q, _hip.ptr(noise), st)


def sweep_accumulates_kinv():
    """PLMC_KINV_IN_SWEEP=1: the sweep accumulates Khat^-1 = W^T W group by group (plmc_potrf with_inverse = 2; fp32: on the
    split engine from the planes of the group's rows of W, on a low-priority stream of its own) and the gradient is one
    HBM-bound pass over it (plmc_grad_tiles).  Default: one fused K^-1 + gradient kernel behind the sweep (plmc_kinv_grad)
    on the gradient stream.  Measured on MI355X (n = 8192, fp32; ms/step fused vs in-sweep), round 3 / split engine:
    q = 8 18.7 / 19.4, q = 2 6.26 / 6.24, q = 1 4.77 / 4.78 (round 2 / fp32 MFMA engine: 37.4 / 37.4, 11.2 / 11.7,
    7.6 / 8.3).  The accumulation does not hide: W is lower triangular, so (g + 1)^2 / 204 of the work belongs to group g --
    54 % of it only becomes available with the last two of eight groups -- and until then the bulk stream is busy back to
    back anyway (profiles/r03_sweep_phases_q8_kinv_in_sweep.txt); beside the chain it slows the chain.  It stays an
    option."""
    return os.environ.get("PLMC_KINV_IN_SWEEP", "0") == "1"


def factorize_checked(kind, X, ell, oscale, noise, rhs, ws, Xs=None):
    """factorize + the jitter ladder of gpytorch's psd_safe_cholesky [gpytorch-knowledge]:
    on a non-PD pivot retry with noise + jitter * 10^k (jitter 1e-6 fp32 / 1e-8 fp64) up to
    settings.cholesky_max_tries, warning each time; raise if still not PD.
    (reference call sites: experiments.py:265, projected_lmc.py:416,649)."""
    factorize(kind, X, ell, oscale, noise, rhs, ws, Xs)
    if not settings.check_cholesky.on():
        return 0.0
    info = ws.info.cpu()
    _check_chain_abort(info)
    if not bool(info.any()):
        return 0.0
    base = settings.cholesky_jitter.value(ws.dtype)
    tries = settings.cholesky_max_tries.value()
    for i in range(tries):
        jit = base * (10 ** i)
        warnings.warn("A not p.d., added jitter of %.1e to the diagonal" % jit, RuntimeWarning)
        factorize(kind, X, ell, oscale, noise + jit, rhs, ws, Xs)
        info = ws.info.cpu()
        if not bool(info.any()):
            return jit
    raise RuntimeError("Matrix not positive definite after repeatedly adding jitter up to %.1e "
                       "(first failing pivot per latent: %s)" % (jit, info.tolist()))


INFO_CHAIN_ABORT = 0x7ffffff0      # csrc/diag_block.hpp: the sweep's resident chain kernel gave up a bounded wait (never a pivot index)


def _check_chain_abort(info_host):
    if bool((info_host == INFO_CHAIN_ABORT).any()):
        raise RuntimeError("projectedlmc: the resident chain kernel of the blocked sweep timed out waiting for another workgroup "
                           "(internal error -- not a property of the matrix); PLMC_CHAIN=0 selects the launch-per-step chain")


class deferred_pivot_checks:
    """Context manager for a caller that can redo its whole forward pass: inside it the exact log-prob does not wait
    for the pivot check of its factorisation (the host goes on queueing the rest of the forward pass while the sweep
    runs); `failed()` after the block waits for the checks.  `jitter` is added to the noise of every factorisation
    inside the block -- the caller's retry ladder (ProjectedLMCmll.forward) plays psd_safe_cholesky's."""
    current = None

    def __init__(self, jitter=0.0):
        self.jitter = float(jitter)
        self.pending = []
        self.first_bad = None

    def __enter__(self):
        self._outer = deferred_pivot_checks.current
        deferred_pivot_checks.current = self
        return self

    def __exit__(self, *exc):
        deferred_pivot_checks.current = self._outer
        return False

    def failed(self):
        bad = [i for i in self.pending if i.failed()]
        self.first_bad = bad[0].host.tolist() if bad else None
        return bool(bad)


class _DeferredInfo:
    """Pivot check of a factorisation without stalling the stream: `info` is copied to pinned host memory right
    behind the sweep and looked at only after the kernels that follow it have been queued, so the GPU runs
    from the sweep straight into them while the host waits for the copy (not for those kernels)."""

    def __init__(self, ws):
        if getattr(ws, "info_host", None) is None:
            ws.info_host = torch.empty(ws.info.shape, dtype=ws.info.dtype, pin_memory=True)
        self.host = ws.info_host
        self.host.copy_(ws.info, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(ws.device))

    def failed(self):
        self.event.synchronize()
        _check_chain_abort(self.host)
        return bool(self.host.any())


class ExactLatentLogProb(torch.autograd.Function):
    """log N(y_i; 0, os_i k(X,X; ell_i) + noise_i I) for a batch of q independent GPs, with the
    analytic gradient computed in the same pass.

    forward(X (n,d), ell (q,d), oscale (q)|None, noise (q), y (q,n), kind) -> (q,)
    forward(..., kind, table) with table (q, d+2) float64: the hyper-parameter gradients are written into `table` and
    NOT returned by this node's backward -- the two-node form exact_latent_log_prob builds when a gradient stream
    exists (see _HyperGrad).
    """

    @staticmethod
    def forward(ctx, X, ell, oscale, noise, y, kind, table=None):
        _hip.require_device(X, ell, noise, y)
        L = _hip.lib()
        dt, dev = y.dtype, y.device
        if dt not in (torch.float32, torch.float64):
            raise TypeError("projectedlmc hot path supports float32 and float64 tensors")
        q, n = y.shape
        d = X.shape[1]
        if d > L.cdll.plmc_max_dim():
            raise ValueError("input dimension %d exceeds plmc_max_dim()=%d" % (d, L.cdll.plmc_max_dim()))
        need_grad = any(ctx.needs_input_grad[1:5]) or table is not None
        Xc, ellc, osc, nzc, yc = (_contig(t, dt) for t in (X, ell, oscale, noise, y))
        ws = get_workspace(n, q, 1, dt, dev, need_grad)
        st = _hip.stream_ptr(dev)
        grad = table if table is not None else (torch.empty(q, d + 2, dtype=torch.float64, device=dev) if need_grad else None)
        check = settings.check_cholesky.on()

        kacc = need_grad and sweep_accumulates_kinv()

        def enqueue(noise_eff):
            """factorisation + everything that consumes it; returns (logp, deferred pivot check)."""
            factorize(kind, Xc, ellc, osc, noise_eff, yc.reshape(q, 1, n), ws, kacc=kacc)
            # (the copy of `info` stays right behind the sweep: the late pivot check of a training step waits for it, and with one
            # latent per rank the host, which then still has the optimiser step and the next projection to queue, is nearly critical)
            info = _DeferredInfo(ws) if check else None
            L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z),
                   _hip.ptr(ws.quad), q, st)
            if need_grad:
                L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z),
                       _hip.ptr(ws.alpha), q, st)
                gs = grad_stream(dev)
                gst = st
                if gs is not None:
                    gs.wait_stream(torch.cuda.current_stream(dev))
                    gst = _hip.stream_handle(gs, dev)
                    for t in (grad, Xc, ellc, osc, noise_eff):
                        if t is not None:
                            t.record_stream(gs)
                if kacc:
                    L.call("plmc_grad_tiles", dt, _hip.KIND[kind], _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, _hip.ptr(ws.Vd),
                           _hip.ptr(ws.alpha), _hip.ptr(Xc), n, d, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(grad), None,
                           _hip.ptr(ws.partials), q, gst)
                else:
                    L.call("plmc_kinv_grad_vd", dt, _hip.KIND[kind], _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW,
                           _hip.ptr(ws.alpha), _hip.ptr(Xc), n, d, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(grad),
                           None, 0, 0, None, _hip.ptr(ws.partials), q, _hip.ptr(noise_eff), _hip.ptr(ws.Vd), gst)
                if gs is not None:
                    ws.pending = torch.cuda.Event()
                    ws.pending.record(gs)
            # (the three small kernels of this line are queued BEHIND the gradient kernel's launch: in front of plmc_wt_matvec they
            # sat on the serial path sweep -> alpha -> gradient kernel)
            lp = -0.5 * (ws.quad + ws.logdet + n * LOG2PI)
            return lp, info

        # jitter ladder of gpytorch's psd_safe_cholesky [gpytorch-knowledge] (see factorize_checked)
        dc = deferred_pivot_checks.current if check else None
        if dc is not None:                 # the caller owns the ladder and looks at the check after its forward pass
            logp, info = enqueue(nzc + dc.jitter if dc.jitter > 0.0 else nzc)
            dc.pending.append(info)
            jit, check = dc.jitter, False
        else:
            logp, info = enqueue(nzc)
            jit = 0.0
        if check and info.failed():
            base, tries = settings.cholesky_jitter.value(dt), settings.cholesky_max_tries.value()
            for i in range(tries):
                jit = base * (10 ** i)
                warnings.warn("A not p.d., added jitter of %.1e to the diagonal" % jit, RuntimeWarning)
                ws = get_workspace(n, q, 1, dt, dev, need_grad)      # waits for the failed attempt's gradient kernel
                logp, info = enqueue(nzc + jit)
                if not info.failed():
                    break
            else:
                raise RuntimeError("Matrix not positive definite after repeatedly adding jitter up to %.1e "
                                   "(first failing pivot per latent: %s)" % (jit, info.host.tolist()))
        if need_grad:
            ctx.save_for_backward(grad, ws.alpha[:, :n].clone())
        ctx.grad_ready = getattr(ws, "pending", None) if need_grad else None
        ctx.d = d
        ctx.has_os = oscale is not None
        ctx.jitter = jit
        ctx.table_out = table is not None
        return logp.to(dt)

    @staticmethod
    def backward(ctx, gout):
        grad, alpha = ctx.saved_tensors
        dt = alpha.dtype
        if ctx.table_out:                  # the hyper-parameter gradients flow through _HyperGrad
            return None, None, None, None, -(gout[:, None].to(dt) * alpha), None, None
        if ctx.grad_ready is not None:
            torch.cuda.current_stream(alpha.device).wait_event(ctx.grad_ready)
        d = ctx.d
        g64 = gout.to(torch.float64)
        g_ell = g64[:, None] * grad[:, :d]
        g_noise = g64 * grad[:, d]
        g_os = g64 * grad[:, d + 1] if ctx.has_os else None
        g_y = -(gout[:, None].to(dt) * alpha)
        return None, g_ell, g_os, g_noise, g_y, None, None


class _HyperGrad(torch.autograd.Function):
    """Second autograd node of the exact log-prob: contributes 0 to the value and carries d logp / d(ell, oscale,
    noise) from the gradient table.  It is applied with the gradient stream current, so autograd runs its backward on
    that stream -- behind the K^-1 + gradient kernel in stream order, no host wait -- and makes the caller's stream wait
    only where the first consumer of these gradients runs (the constraint transforms of the kernel parameters, created
    early in the forward pass and therefore late in the backward pass).  The backward of the projection (d logp / dy
    -> mixing matrix, ~50 launch-bound kernels) is queued ahead of that wait and runs beside the gradient kernel."""

    @staticmethod
    def forward(ctx, ell, oscale, noise, table, zero):
        ctx.save_for_backward(table)
        ctx.d = ell.shape[1]
        ctx.has_os = oscale is not None
        return zero.detach()

    @staticmethod
    def backward(ctx, gout):
        (table,) = ctx.saved_tensors
        d = ctx.d
        g = (table * gout[:, None]).to(gout.dtype)          # fp64 product, one cast; the three gradients are views
        return g[:, :d], (g[:, d + 1] if ctx.has_os else None), g[:, d], None, None


_zeros = {}


class HyperGradHandle:
    __slots__ = ("table", "hz")


def prepare_hyper_grad(ell, oscale, noise):
    """Create the hyper-parameter gradient node (_HyperGrad) NOW and return a handle for exact_latent_log_prob, or
    None when there is nothing to do (no gradient stream, nothing requires grad, host tensors).
    Autograd runs backward nodes in reverse creation order, and the caller's stream is made to wait for the gradient
    stream as soon as this node has run; a caller that creates it BEFORE the graph that produces y (the projection
    of ProjectedLMCmll: QR, triangular solve, GEMMs) gets that graph's backward queued ahead of the wait."""
    hyper = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (ell, oscale, noise))
    gs = grad_stream(ell.device) if (hyper and ell.is_cuda) else None
    if gs is None:
        return None
    q, d = ell.shape
    key = (ell.device.index, ell.dtype, q)
    zero = _zeros.get(key)
    if zero is None:
        zero = _zeros[key] = torch.zeros(q, dtype=ell.dtype, device=ell.device)
        torch.cuda.current_stream(ell.device).synchronize()
    h = HyperGradHandle()
    h.table = torch.empty(q, d + 2, dtype=torch.float64, device=ell.device)
    with torch.cuda.stream(gs):
        h.hz = _HyperGrad.apply(ell, oscale, noise, h.table, zero)
    return h


def exact_latent_log_prob(kind, X, ell, oscale, noise, y, hyper=None):
    if hyper is None:
        hyper = prepare_hyper_grad(ell, oscale, noise)
    if hyper is None:
        return ExactLatentLogProb.apply(X, ell, oscale, noise, y, kind)
    det = lambda t: None if t is None else t.detach()
    lp = ExactLatentLogProb.apply(X, det(ell), det(oscale), det(noise), y, kind, hyper.table)
    return lp + hyper.hz


def exact_loo(kind, X, ell, oscale, noise, y):
    """Leave-one-out moments of q independent exact GPs sharing the factorisation the MLL uses
    (reference: compute_loo, projected_lmc.py:371-436, 1108-1119):
        sigma2_i = 1 / [Khat^-1]_ii ,   (y - mu_loo)_i = [Khat^-1 y]_i * sigma2_i .
    diag(Khat^-1) is a by-product of the fused K^-1 kernel (`kinv_diag` output).  Returns (q,n),(q,n)."""
    _hip.require_device(X, ell, noise, y)
    L = _hip.lib()
    dt, dev = y.dtype, y.device
    q, n = y.shape
    d = X.shape[1]
    Xc, ellc, osc, nzc, yc = (_contig(t, dt) for t in (X, ell, oscale, noise, y))
    ws = get_workspace(n, q, 1, dt, dev, True)
    st = _hip.stream_ptr(dev)
    factorize_checked(kind, Xc, ellc, osc, nzc, yc.reshape(q, 1, n), ws)
    L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z),
           _hip.ptr(ws.quad), q, st)
    L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z),
           _hip.ptr(ws.alpha), q, st)
    grad = torch.empty(q, d + 2, dtype=torch.float64, device=dev)
    kd = torch.empty(q, ws.n_pad, dtype=dt, device=dev)
    L.call("plmc_kinv_grad_vd", dt, _hip.KIND[kind], _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.alpha),
           _hip.ptr(Xc), n, d, _hip.ptr(ellc), _hip.ptr(osc), _hip.ptr(grad), None, 0, 0, _hip.ptr(kd),
           _hip.ptr(ws.partials), q, _hip.ptr(nzc), _hip.ptr(ws.Vd), st)
    sigma2 = 1.0 / kd[:, :n]
    return sigma2, ws.alpha[:, :n] * sigma2


class PosteriorCache:
    """What gpytorch's prediction strategy keeps between eval-mode calls of one model (projected_lmc.py:1133-1134,
    experiments.py:316-331 predicts batch by batch): the factorisation of the training covariance.  Here: a workspace of
    its own holding U, the inverse factor W and room for the augmented columns [y | K*^T].  The first call does the whole
    augmented sweep; while `key` (the model's parameter / data versions) is unchanged, a later call only rewrites the
    augmented columns and forward-substitutes them (plmc_potrs_aug): n^2 n* flops instead of n^3 / 3 + n^2 n*."""

    def __init__(self):
        self.key, self.ws = None, None
        self.seen = None                 # state key of the last uncached call (the cache is built on the SECOND call with one key)
        self.hits = self.misses = 0

    def drop(self):
        self.key, self.ws = None, None
        self.seen = None


def model_state_key(module, *tensors):
    """Cache key of an eval-mode model: identity and version counter of every parameter / buffer and of the given tensors
    (an optimiser step, load_state_dict or an in-place edit bumps a version; a new tensor has a new identity)."""
    items = [(id(t), t._version) for t in list(module.parameters()) + list(module.buffers())]
    items += [(id(t), t._version, tuple(t.shape)) for t in tensors if t is not None]
    return tuple(items)


def exact_posterior(kind, X, ell, oscale, noise, y, Xs, full_cov=False, cache=None, key=None):
    """Posterior of q zero-mean GPs at Xs from ONE augmented factorization [Khat | y | K*^T]:
    with v = U^-T k*, z = U^-T y:  mean = v^T z,  cov = K** - v^T v.
    (ExactGPModel.__call__ in eval mode -> gpytorch DefaultPredictionStrategy; reached from
    projected_lmc.py:1134.)  Returns (mean (q,ns), var (q,ns) | cov (q,ns,ns)).
    cache / key: a PosteriorCache owned by the calling model and its state key -- see PosteriorCache.  The cache is built
    LAZILY (settings.prediction_cache "lazy", the default): the first call with a given key runs the plain augmented sweep
    (n^3 / 3 flops, a workspace from the shared pool) and only remembers the key; the second call with the same key pays for the
    sweep with the inverse factor and the kept planes (2 n^3 / 3, 13 GB at n = 8192, q = 8; ~50 GB per latent at n = 44 484) that
    every later call then reuses -- a one-shot prediction neither slows down nor holds that memory.  "eager" builds it
    on the first call, "off" never."""
    _hip.require_device(X, ell, noise, y, Xs)
    L = _hip.lib()
    dt, dev = y.dtype, y.device
    q, n = y.shape
    ns = Xs.shape[0]
    Xc, Xsc, ellc, osc, nzc = (_contig(t, dt) for t in (X, Xs, ell, oscale, noise))
    yc = _contig(y).reshape(q, 1, n)
    st = _hip.stream_ptr(dev)
    mode = settings.prediction_cache.value() if cache is not None else "off"
    hit = False
    if cache is not None and mode != "off":
        ws = cache.ws
        hit = (ws is not None and cache.key == key and key is not None and ws.dtype == dt and ws.device == dev and ws.n == n and ws.q == q
               and ws.naug >= 1 + ns)
    # build: eager mode; lazy mode on the second call with one key; or the cached state is the same and only its capacity is short
    same_state = cache is not None and cache.ws is not None and key is not None and cache.key == key
    build = cache is not None and mode != "off" and not hit and (mode == "eager" or same_state or (key is not None and cache.seen == key))
    if not hit and not build:
        if cache is not None and mode != "off":
            cache.drop()                                   # (a stale factorisation of another state goes now, not at the next build)
            cache.seen = key
            cache.misses += 1
        ws = get_workspace(n, q, 1 + ns, dt, dev, False)
        factorize_checked(kind, Xc, ellc, osc, nzc, yc, ws, Xs=Xsc)
    else:
        if hit:
            # new right-hand sides into the factorised buffer: y (column 0 shares its tile with the first test points, so it
            # is rewritten raw as well), K*^T behind it; then the forward substitution of those columns only
            cache.hits += 1
            k = _hip.KIND[kind]
            L.call("plmc_write_rhs", dt, _hip.ptr(yc), 1, n, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad, q, st)
            L.call("plmc_assemble_cross", dt, k, _hip.ptr(Xc), n, _hip.ptr(Xsc), ns, Xc.shape[1], _hip.ptr(ellc), _hip.ptr(osc),
                   _hip.ptr(ws.A), ws.lda, ws.strideA, ws.n_pad + 1, ws.n_pad, q, st)
            if ws.keep_planes:
                L.call("plmc_potrs_aug_kept", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, 1 + ns, ws.wcol0, ws.strideA, _hip.ptr(ws.Vd), q,
                       _hip.ptr(nzc), st)
            else:
                L.call("plmc_potrs_aug", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, 1 + ns, ws.wcol0, ws.strideA, _hip.ptr(ws.Vd), q, st)
        else:
            cache.misses += 1
            cache.drop()
            ws = Workspace(n, q, 1 + ns, dt, dev, with_inverse=True, keep_planes=True)
            factorize_checked(kind, Xc, ellc, osc, nzc, yc, ws, Xs=Xsc)
            cache.key, cache.ws = key, ws
    # mean = V^T z and |v|^2 per test point: one pass over the augmented columns (plmc_posterior_moments)
    mean = torch.empty(q, ns, dtype=dt, device=dev)
    vsq = torch.empty(q, ns, dtype=dt, device=dev)
    L.call("plmc_posterior_moments", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, ns, _hip.ptr(mean), _hip.ptr(vsq), q, st)
    if full_cov:
        V = ws.A[:, :, ws.n_pad + 1:ws.n_pad + 1 + ns]             # (q, n_pad, ns) strided view, K-major
        Kss = dense_cross(kind, Xsc, Xsc, ellc, osc)
        from ._dense import gemm_tn
        cov = Kss - gemm_tn(V, V)                                   # V^T V on the library's tile engine (plmc_gemm_tn)
        return mean, cov
    from .kernels import prior_diagonal
    var = prior_diagonal(kind, Xsc, osc, q) - vsq                      # k(x*, x*) = 1 for the stationary kinds
    return mean, var


def mix_posterior(mean_lat, var_lat, Ht, eps=0.0):
    """Task-space moments of the projected model (projected_lmc.py:1144, :1152): mean (ns, p) = mean_lat^T Ht,
    var (ns, p) = var_lat^T Ht^2 + eps, from latent moments (q, ns) and the mixing matrix Ht (q, p) -- plmc_mix_posterior."""
    _hip.require_device(mean_lat, var_lat, Ht)
    L = _hip.lib()
    dt, dev = mean_lat.dtype, mean_lat.device
    q, ns = mean_lat.shape
    p = Ht.shape[1]
    ml, vl, H = (_contig(t, dt) for t in (mean_lat, var_lat, Ht))
    mean = torch.empty(ns, p, dtype=dt, device=dev)
    var = torch.empty(ns, p, dtype=dt, device=dev)
    L.call("plmc_mix_posterior", dt, _hip.ptr(ml), _hip.ptr(vl), _hip.ptr(H), q, ns, p, float(eps), _hip.ptr(mean), _hip.ptr(var),
           _hip.stream_ptr(dev))
    return mean, var


def dense_cross(kind, X1, X2, ell, oscale):
    """Dense k(X1, X2) per latent via the cross-assembly kernel: (q, n1, n2)."""
    L = _hip.lib()
    dt, dev = X1.dtype, X1.device
    q = ell.shape[0]
    n1, n2, d = X1.shape[0], X2.shape[0], X1.shape[1]
    out = torch.empty(q, n1, n2, dtype=dt, device=dev)
    L.call("plmc_assemble_cross", dt, _hip.KIND[kind], _hip.ptr(X1.contiguous()), n1, _hip.ptr(X2.contiguous()), n2, d,
           _hip.ptr(ell.contiguous()), _hip.ptr(None if oscale is None else oscale.contiguous()), _hip.ptr(out),
           n2, n1 * n2, 0, n1, q, _hip.stream_ptr(dev))
    return out
