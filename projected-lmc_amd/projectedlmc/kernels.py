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
    def _ell(self, d, like=None):
        """(q, d) lengthscales (q = prod(batch_shape) or 1)."""
        ell = self.lengthscale.reshape(-1, self.lengthscale.shape[-1])
        return ell.expand(ell.shape[0], d) if ell.shape[-1] != d else ell

    def _pieces(self, d, like=None):
        return self.kind, self._ell(d, like), None

    def select(self, x):
        if self.active_dims is not None and len(self.active_dims) != x.shape[-1]:
            return x[..., list(self.active_dims)]
        return x

    def forward(self, x1, x2=None, **params):
        x1 = self.select(x1)
        x2 = x1 if x2 is None else self.select(x2)
        kind, ell, osc = self._pieces(x1.shape[-1], x1)
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


class SplineKernel(Kernel):
    """The reference's SplineKernel (projected_lmc.py:26-36): k(x, x') = prod_k [1 + m M + m^2 (M - m / 3) / 2] with
    m = min(x_k, x'_k), M = max(x_k, x'_k); no lengthscale, k(x, x) = prod_k (1 + x_k^2 + x_k^3 / 3).  On the HIP path it
    is kernel kind "spline" of the assembly / cross / gradient kernels (exact GP and projected models; the dense-LMC and
    variational engines take the stationary kinds only)."""
    has_lengthscale = False
    kind = "spline"
    is_stationary = True                      # as declared by the reference

    def _ell(self, d, like=None):
        q = max(1, int(self.batch_shape.numel()))
        if like is None:
            return torch.ones(q, d)
        return torch.ones(q, d, dtype=like.dtype, device=like.device)


def prior_diagonal(kind, x, oscale, q):
    """k(x, x) per latent, (q, n): 1 for the stationary kinds, prod_k (1 + x_k^2 + x_k^3 / 3) for the spline kernel."""
    if kind == "spline":
        dg = (1 + x ** 2 + x ** 3 / 3).prod(dim=-1).reshape(1, -1).expand(q, -1)
    else:
        dg = torch.ones(q, x.shape[-2], dtype=x.dtype, device=x.device)
    return dg if oscale is None else dg * oscale.reshape(-1, 1)


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

    def _pieces(self, d, like=None):
        kind, ell, _ = self.base_kernel._pieces(d, like if like is not None else self.raw_outputscale)
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
        dg = prior_diagonal(self.kind, self.x1.to(self.ell.dtype), self.oscale, q)
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


# ------------------------------------------------------------------------------------------------
# Multitask (Kronecker) kernels of the exact LMC / ICM models (projected_lmc.py:462-466).
class IndexKernel(torch.nn.Module):
    """Task covariance B = F F^T + diag(softplus(raw_var)) [gpytorch-knowledge: IndexKernel;
    covar_factor (p x rank) and raw_var (p) are randn-initialised]."""

    def __init__(self, num_tasks, rank=1, **kwargs):
        super().__init__()
        self.num_tasks, self.rank = num_tasks, rank
        self.register_parameter("covar_factor", torch.nn.Parameter(torch.randn(num_tasks, rank)))
        self.register_parameter("raw_var", torch.nn.Parameter(torch.randn(num_tasks)))
        self.raw_var_constraint = Positive()

    @property
    def var(self):
        return self.raw_var_constraint.transform(self.raw_var)

    @property
    def covar_matrix(self):
        F = self.covar_factor
        return F @ F.transpose(-1, -2) + torch.diag_embed(self.var.to(F.dtype))


class MultitaskKernel(Kernel):
    """K_data(x,x') (x) B (data-major interleaving) [gpytorch-knowledge: MultitaskKernel]."""

    def __init__(self, data_covar_module, num_tasks, rank=1, **kwargs):
        super().__init__()
        self.data_covar_module = data_covar_module
        self.task_covar_module = IndexKernel(num_tasks=num_tasks, rank=rank)
        self.num_tasks = num_tasks

    def _lmc_pieces(self, d):
        kind, ell, osc = self.data_covar_module._pieces(d)
        return kind, ell[:1], None if osc is None else osc[:1], self.task_covar_module.covar_matrix.unsqueeze(0)

    def forward(self, x1, x2=None, **params):
        x1 = self.data_covar_module.select(x1)
        kind, ell, osc, B = self._lmc_pieces(x1.shape[-1])
        return LazyLmcKernel(kind, x1, ell, osc, B)


class LCMKernel(Kernel):
    """sum_i K_i (x) B_i, one MultitaskKernel per latent [gpytorch-knowledge: LCMKernel]."""

    def __init__(self, base_kernels, num_tasks, rank=1, **kwargs):
        super().__init__()
        self.covar_module_list = torch.nn.ModuleList(
            [MultitaskKernel(b, num_tasks=num_tasks, rank=rank) for b in base_kernels])
        self.num_tasks = num_tasks

    def forward(self, x1, x2=None, **params):
        x1 = self.covar_module_list[0].data_covar_module.select(x1)
        pieces = [m._lmc_pieces(x1.shape[-1]) for m in self.covar_module_list]
        kind = pieces[0][0]
        ell = torch.cat([p_[1] for p_ in pieces], 0)
        osc = None if pieces[0][2] is None else torch.cat([p_[2] for p_ in pieces], 0)
        B = torch.cat([p_[3] for p_ in pieces], 0)
        return LazyLmcKernel(kind, x1, ell, osc, B)


class LazyLmcKernel:
    """Un-evaluated sum_i os_i k(X,X; ell_i) (x) B_i (+ I (x) Sigma once a multitask likelihood was
    applied); consumed by the HIP LMC engine."""

    def __init__(self, kind, x, ell, oscale, B, task_noise=None):
        self.kind, self.x, self.ell, self.oscale, self.B, self.task_noise = kind, x, ell, oscale, B, task_noise

    @property
    def shape(self):
        N = self.x.shape[-2] * self.B.shape[-1]
        return torch.Size([N, N])

    def add_task_noise(self, Sigma):
        tn = Sigma if self.task_noise is None else self.task_noise + Sigma
        return LazyLmcKernel(self.kind, self.x, self.ell, self.oscale, self.B, tn)

    def diagonal(self, *a, **k):
        q = self.ell.shape[0]
        os_ = torch.ones(q, dtype=self.B.dtype, device=self.B.device) if self.oscale is None else self.oscale
        dg = (os_[:, None] * torch.diagonal(self.B, dim1=-2, dim2=-1)).sum(0)
        if self.task_noise is not None:
            dg = dg + torch.diagonal(self.task_noise)
        return dg.repeat(self.x.shape[-2])

    def log_prob_flat(self, diff):
        from . import _lmc_engine
        if self.task_noise is None:
            raise RuntimeError("log_prob of a noise-free LMC prior: apply the multitask likelihood first")
        return _lmc_engine.lmc_exact_log_prob(self.kind, self.x, self.ell, self.oscale, self.B, self.task_noise,
                                              diff.reshape(-1))

    def evaluate(self):
        """Dense (np x np) matrix via the HIP cross kernel (small cases / inspection only)."""
        from . import _hip
        L = _hip.lib()
        x = self.x.contiguous()
        n, d = x.shape
        q, p = self.B.shape[0], self.B.shape[-1]
        dt, dev = self.B.dtype, self.B.device
        out = torch.empty(n * p, n * p, dtype=dt, device=dev)
        L.call("plmc_lmc_cross", dt, _hip.KIND[self.kind], _hip.ptr(x.to(dt)), n, _hip.ptr(x.to(dt)), n, d, p, q,
               _hip.ptr(self.ell.detach().to(dt).contiguous()),
               _hip.ptr(None if self.oscale is None else self.oscale.detach().to(dt).contiguous()),
               _hip.ptr(self.B.detach().contiguous()), _hip.ptr(out), n * p, 0, n * p, _hip.stream_ptr(dev))
        if self.task_noise is not None:
            out = out + torch.kron(torch.eye(n, dtype=dt, device=dev), self.task_noise.detach())
        return out

    to_dense = evaluate
