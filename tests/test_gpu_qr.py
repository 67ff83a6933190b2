"""Single-launch small QR (csrc/qr_small.hip) against torch.linalg.qr on the CPU in fp64 -- the call the reference
makes (LMCMixingMatrix.QR, projected_lmc.py:864-875) -- values, LAPACK sign convention, and the backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from projectedlmc import _qr  # noqa: E402


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(16, 16), (16, 8), (7, 3), (1, 1), (5, 5), (64, 64), (64, 1), (33, 32)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_qr_matches_lapack(shape, dtype):
    g = torch.Generator().manual_seed(shape[0] * 100 + shape[1])
    A = torch.randn(*shape, generator=g, dtype=torch.float64)
    Qr, Rr = torch.linalg.qr(A.to(dtype).double())
    Q, R = _qr.qr(A.to(_dev(), dtype))
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert Q.shape == Qr.shape and R.shape == Rr.shape
    assert torch.allclose(Q.double().cpu(), Qr, atol=tol, rtol=0), (Q.double().cpu() - Qr).abs().max()
    assert torch.allclose(R.double().cpu(), Rr, atol=tol * max(1.0, float(Rr.abs().max())), rtol=0)
    assert float(R.tril(-1).abs().max()) == 0.0 if shape[1] > 1 else True


def test_qr_zero_tail_column_keeps_sign():
    """xGEQR2: a column that is already zero below the diagonal gets tau = 0 and R_kk keeps its sign (the last column
    of a square matrix always does)."""
    A = torch.tensor([[2.0, 1.0, 3.0], [0.0, -4.0, 1.0], [0.0, 0.0, -5.0]], dtype=torch.float64)
    Qr, Rr = torch.linalg.qr(A)
    Q, R = _qr.qr(A.to(_dev()))
    assert torch.allclose(Q.cpu(), Qr, atol=1e-13) and torch.allclose(R.cpu(), Rr, atol=1e-13)


@pytest.mark.parametrize("shape", [(16, 16), (16, 8), (6, 2)])
def test_qr_backward_matches_torch(shape):
    g = torch.Generator().manual_seed(7)
    A0 = torch.randn(*shape, generator=g, dtype=torch.float64)
    cQ = torch.randn(*shape, generator=g, dtype=torch.float64)
    cR = torch.randn(shape[1], shape[1], generator=g, dtype=torch.float64).triu()
    Ar = A0.clone().requires_grad_()
    Qr, Rr = torch.linalg.qr(Ar)
    ((Qr * cQ).sum() + (Rr * cR).sum()).backward()
    Ad = A0.to(_dev()).requires_grad_()
    Q, R = _qr.qr(Ad)
    ((Q * cQ.to(_dev())).sum() + (R * cR.to(_dev())).sum()).backward()
    assert torch.allclose(Ad.grad.cpu(), Ar.grad, atol=1e-10, rtol=1e-10), (Ad.grad.cpu() - Ar.grad).abs().max()
    # only one of the two outputs used
    Ad2 = A0.to(_dev()).requires_grad_()
    _qr.qr(Ad2)[1].diagonal().abs().log().sum().backward()
    Ar2 = A0.clone().requires_grad_()
    torch.linalg.qr(Ar2)[1].diagonal().abs().log().sum().backward()
    assert torch.allclose(Ad2.grad.cpu(), Ar2.grad, atol=1e-10, rtol=1e-10)


def test_qr_larger_than_kernel_limit_uses_device_torch():
    A = torch.randn(80, 70, dtype=torch.float64, device=_dev())
    Q, R = _qr.qr(A)
    assert torch.allclose(Q @ R, A, atol=1e-12)
