"""Kernel modules: parameter containers with gpytorch's constructor / attribute surface
(`ard_num_dims`, `active_dims`, `batch_shape`, `lengthscale_prior`, `.lengthscale`,
`ScaleKernel.outputscale`, `.base_kernel`) as used by handle_covar_ (projected_lmc.py:151-179).

Calling a kernel does NOT build an n x n tensor: it returns a `LazyKernel` descriptor that the
HIP engine consumes (fused assembly inside the factorization).  `.evaluate()` / `.to_dense()`
materialise through the HIP cross-assembly kernel when a dense matrix is explicitly asked for.
"""
import torch

from .constraints import Positive

_MATERN_KIND = {0.5: "matern12", 1.5: "matern32", 2.5: "matern52"}


class Kernel(torch.nn.Module):
    has_lengthscale = False
    kind = None

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size(), active_dims=None, lengthscale_prior=None,
                 lengthscale_constraint=None, **kwargs):
        super().__init__()
        self.ard_num_dims = ard_num_dims
        self.batch_shape = torch.Size(batch_shape)
        self.active_dims = None if active_dims is None else tuple(int(a) for a in active_dims)
        self.lengthscale_prior = lengthscale_prior
        if self.has_lengthscale:
            nd = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, nd)))
            self.raw_lengthscale_constraint = lengthscale_constraint or Positive()

    @property
    def lengthscale(self):
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale) if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        value = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype, device=self.raw_lengthscale.device)
        raw = self.raw_lengthscale_constraint.inverse_transform(value)
        with torch.no_grad():
            self.raw_lengthscale.copy_(raw.expand_as(self.raw_lengthscale))

    # -- descriptor pieces consumed by the engine
    def _ell(self, d):
        """(q, d) lengthscales (q = prod(batch_shape) or 1)."""
        ell = self.lengthscale.reshape(-1, self.lengthscale.shape[-1])
        return ell.expand(ell.shape[0], d) if ell.shape[-1] != d else ell

    def _pieces(self, d):
        return self.kind, self._ell(d), None

    def select(self, x):
        if self.active_dims is not None and len(self.active_dims) != x.shape[-1]:
            return x[..., list(self.active_dims)]
        return x

    def forward(self, x1, x2=None, **params):
        x1 = self.select(x1)
        x2 = x1 if x2 is None else self.select(x2)
        kind, ell, osc = self._pieces(x1.shape[-1])
        return LazyKernel(kind, x1, x2, ell, osc, self.batch_shape)


class RBFKernel(Kernel):
    has_lengthscale = True
    kind = "rbf"


class MaternKernel(Kernel):
    has_lengthscale = True

    def __init__(self, nu=2.5, **kwargs):
        if nu not in _MATERN_KIND:
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        super().__init__(**kwargs)
        self.nu = nu
        self.kind = _MATERN_KIND[nu]


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, batch_shape=torch.Size(),
                 **kwargs):
        super().__init__(batch_shape=batch_shape, active_dims=base_kernel.active_dims)
        self.base_kernel = base_kernel
        self.register_parameter("raw_outputscale", torch.nn.Parameter(torch.zeros(*self.batch_shape)))
        self.raw_outputscale_constraint = outputscale_constraint or Positive()

    @property
    def outputscale(self):
        return self.raw_outputscale_constraint.transform(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        value = torch.as_tensor(value, dtype=self.raw_outputscale.dtype, device=self.raw_outputscale.device)
        with torch.no_grad():
            self.raw_outputscale.copy_(self.raw_outputscale_constraint.inverse_transform(value).expand_as(self.raw_outputscale))

    def _pieces(self, d):
        kind, ell, _ = self.base_kernel._pieces(d)
        return kind, ell, self.outputscale.reshape(-1)


class LazyKernel:
    """Un-evaluated batched covariance os * k(x1, x2; ell) (+ noise * I once a likelihood was
    applied).  The hot path never materialises it."""

    def __init__(self, kind, x1, x2, ell, oscale, batch_shape, noise=None):
        self.kind, self.x1, self.x2, self.ell, self.oscale, self.noise = kind, x1, x2, ell, oscale, noise
        self.batch_shape = torch.Size(batch_shape)
        self.is_square = x1 is x2

    @property
    def shape(self):
        return torch.Size([*self.batch_shape, self.x1.shape[-2], self.x2.shape[-2]])

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    @property
    def dtype(self):
        return self.x1.dtype

    @property
    def device(self):
        return self.x1.device

    def add_noise(self, noise):
        return LazyKernel(self.kind, self.x1, self.x2, self.ell, self.oscale, self.batch_shape,
                          noise if self.noise is None else self.noise + noise)

    def diagonal(self, *args, **kwargs):
        q = self.ell.shape[0]
        os_ = torch.ones(q, dtype=self.dtype, device=self.device) if self.oscale is None else self.oscale
        dg = os_[:, None].expand(q, self.x1.shape[-2])
        if self.noise is not None:
            dg = dg + self.noise.reshape(-1, 1)
        return dg.reshape(*self.batch_shape, -1)

    def evaluate(self):
        from . import _engine
        K = _engine.dense_cross(self.kind, self.x1.to(self.ell.dtype), self.x2.to(self.ell.dtype), self.ell.detach(),
                                None if self.oscale is None else self.oscale.detach())
        if self.noise is not None and self.is_square:
            K = K + self.noise.detach().reshape(-1, 1, 1) * torch.eye(K.shape[-1], dtype=K.dtype, device=K.device)
        return K.reshape(*self.batch_shape, *K.shape[-2:])

    to_dense = evaluate
    evaluate_kernel = lambda self: self  # noqa: E731  (gpytorch idiom used at projected_lmc.py:368)
