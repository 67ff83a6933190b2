"""Reduced QR of the small mixing matrix as ONE device launch, with the analytic backward.

Reference: `torch.linalg.qr(self.H)` in LMCMixingMatrix.QR, bulk mode (projected_lmc.py:864-875).  On the device
torch runs that as rocSOLVER geqrf + orgqr, ~45 dependent launches per call; `plmc_qr_small` (csrc/qr_small.hip)
does the same Householder factorisation (LAPACK sign convention, so the factors agree with torch's to rounding)
in one.  The backward is the closed form torch uses for mode='reduced', m >= n [torch FunctionsManual
linalg_qr_backward]:      gA = [gQ + Q syminvadj(triu(gR R^T - Q^T gQ))] R^-T,
syminvadj(X) = X + X^T with the diagonal halved -- a handful of p x p torch ops."""
import os

import torch

from . import _hip


class SmallQR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A):
        _hip.require_device(A)
        L = _hip.lib()
        m, n = A.shape
        Ac = A.detach().contiguous()
        Q = torch.empty(m, n, dtype=A.dtype, device=A.device)
        R = torch.empty(n, n, dtype=A.dtype, device=A.device)
        L.call("plmc_qr_small", A.dtype, _hip.ptr(Ac), m, n, n, _hip.ptr(Q), n, _hip.ptr(R), n,
               _hip.stream_ptr(A.device))
        ctx.save_for_backward(Q, R)
        return Q, R

    @staticmethod
    def backward(ctx, gQ, gR):
        Q, R = ctx.saved_tensors
        return qr_backward(Q, R, gQ, gR)


def qr_backward(Q, R, gQ, gR):
    """gA for A = Q R (reduced, m >= n) from the cotangents of Q and R (either may be None)."""
    if gQ is None and gR is None:
        return None
    b = torch.zeros_like(R)
    if gR is not None:
        b = b + gR @ R.mT
    if gQ is not None:
        b = b - Q.mT @ gQ
    b = b.triu()
    b = b + b.mT
    b.diagonal().mul_(0.5)
    b = Q @ b
    if gQ is not None:
        b = b + gQ
    return torch.linalg.solve_triangular(R.mT, b, upper=False, left=False)


def supported(A):
    if os.environ.get("PLMC_SMALL_QR", "1") == "0":      # dev knob: torch.linalg.qr on the device (rocSOLVER)
        return False
    return (A.dim() == 2 and A.is_cuda and A.dtype in (torch.float32, torch.float64)
            and 1 <= A.shape[1] <= A.shape[0] <= _hip.lib().cdll.plmc_qr_max())


def qr(A):
    """(Q, R) = reduced QR of a 2-D device matrix; single-launch kernel when it fits, torch.linalg.qr otherwise."""
    if supported(A):
        return SmallQR.apply(A)
    return torch.linalg.qr(A)
