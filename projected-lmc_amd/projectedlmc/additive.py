"""Additive sub-kernel decompositions, `decomp=[[0,1],[1,2]]` -> k(x) = s1 k1(x0,x1) + s2 k2(x1,x2)
(reference: handle_covar_, projected_lmc.py:131-167).

HIP path: a sum of G scaled ARD kernels on subsets of the input dimensions is the dense LMC
covariance with one task (p = 1, B_g = [[1]]) whose g-th "latent" has an infinite lengthscale on the
dimensions it ignores (1/ell = 0), so assembly, factorisation and all gradients reuse csrc/lmc.hip.
Single-output models only (batch of 1); batched latents with additive kernels are not built."""
import torch

from .kernels import Kernel, ScaleKernel, LazyLmcKernel


class AdditiveKernel(Kernel):
    def __init__(self, *kernels):
        super().__init__()
        self.kernels = torch.nn.ModuleList(kernels)

    def __add__(self, other):
        return AdditiveKernel(*self.kernels, *(other.kernels if isinstance(other, AdditiveKernel) else [other]))

    def select(self, x):
        return x

    def forward(self, x1, x2=None, **params):
        d = x1.shape[-1]
        kinds, ells, oss = [], [], []
        for k in self.kernels:
            base = k.base_kernel if isinstance(k, ScaleKernel) else k
            if base.batch_shape.numel() > 1:
                raise NotImplementedError("additive kernels are supported for single-output models only")
            dims = list(base.active_dims) if base.active_dims is not None else list(range(d))
            ell_g = torch.full((d,), float("inf"), dtype=base.lengthscale.dtype, device=base.lengthscale.device)
            ell_g = ell_g.index_put((torch.tensor(dims, device=ell_g.device),), base.lengthscale.reshape(-1))
            kinds.append(base.kind)
            ells.append(ell_g)
            oss.append(k.outputscale.reshape(-1)[0] if isinstance(k, ScaleKernel) else torch.ones((), dtype=ell_g.dtype, device=ell_g.device))
        if len(set(kinds)) != 1:
            raise NotImplementedError("all sub-kernels of a decomposition must be of the same type")
        G = len(ells)
        B = torch.ones(G, 1, 1, dtype=ells[0].dtype, device=ells[0].device)
        return LazyAdditiveKernel(kinds[0], x1, torch.stack(ells), torch.stack(oss), B)


class LazyAdditiveKernel(LazyLmcKernel):
    """LazyLmcKernel with p = 1 and the single-output hooks (noise instead of task noise)."""

    def add_noise(self, noise):
        return LazyAdditiveKernel(self.kind, self.x, self.ell, self.oscale, self.B,
                                  noise.reshape(1, 1) if self.task_noise is None else self.task_noise + noise.reshape(1, 1))

    @property
    def x1(self):
        return self.x

    def log_prob_batch(self, y):
        return self.log_prob_flat(y.reshape(-1)).reshape(1)

    def posterior(self, y, xs):
        from . import _lmc_engine
        mean, var = _lmc_engine.lmc_posterior(self.kind, self.x, self.ell.detach(), self.oscale.detach(), self.B,
                                              self.task_noise.detach(), y.reshape(-1), xs)
        return mean.reshape(1, -1), var.reshape(1, -1)
