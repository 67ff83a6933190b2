"""Additive sub-kernel decompositions, `decomp=[[0,1],[1,2]]` -> k(x) = s1 k1(x0,x1) + s2 k2(x1,x2)
(reference: handle_covar_, projected_lmc.py:131-167).

HIP path: a sum of G scaled ARD kernels on subsets of the input dimensions is the dense LMC
covariance with one task (p = 1, B_g = [[1]]) whose g-th "latent" has an infinite lengthscale on the
dimensions it ignores (1/ell = 0), so assembly, factorisation and all gradients reuse csrc/lmc.hip.
A batch of q functions (the reference builds the decomposition with batch_shape=[n_funcs], projected_lmc.py:151-167:
batched exact GPs, the latent processes of the projected model) is q such problems, one factorisation each."""
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
        nb = max(int((k.base_kernel if isinstance(k, ScaleKernel) else k).batch_shape.numel()) for k in self.kernels)
        for k in self.kernels:
            base = k.base_kernel if isinstance(k, ScaleKernel) else k
            dims = list(base.active_dims) if base.active_dims is not None else list(range(d))
            ls = base.lengthscale.reshape(-1, len(dims))                       # (batch, |dims|)
            ell_g = torch.full((ls.shape[0], d), float("inf"), dtype=ls.dtype, device=ls.device)
            ell_g = ell_g.index_copy(1, torch.tensor(dims, device=ls.device), ls)
            kinds.append(base.kind)
            ells.append(ell_g.expand(nb, d))
            os_g = k.outputscale.reshape(-1) if isinstance(k, ScaleKernel) else torch.ones(1, dtype=ls.dtype, device=ls.device)
            oss.append(os_g.expand(nb))
        if len(set(kinds)) != 1:
            raise NotImplementedError("all sub-kernels of a decomposition must be of the same type")
        G = len(ells)
        ell, osc = torch.stack(ells, 1), torch.stack(oss, 1)                   # (batch, G, d), (batch, G)
        B = torch.ones(G, 1, 1, dtype=ell.dtype, device=ell.device)
        if nb == 1:
            return LazyAdditiveKernel(kinds[0], x1, ell[0], osc[0], B)
        return LazyBatchedAdditiveKernel([LazyAdditiveKernel(kinds[0], x1, ell[i], osc[i], B) for i in range(nb)])


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


class LazyBatchedAdditiveKernel:
    """q additive kernels on the same inputs (batch_shape = [q]): the hooks of the single-output form, looped --
    every latent is its own dense factorisation (one per latent, as the blocked sweep batches independent matrices;
    the Kronecker-sum assembly kernel evaluates one sum of sub-kernels per launch)."""

    def __init__(self, parts):
        self.parts = parts

    @property
    def kind(self):
        return self.parts[0].kind

    @property
    def x1(self):
        return self.parts[0].x

    @property
    def ell(self):
        return self.parts[0].ell

    @property
    def shape(self):
        n = self.parts[0].x.shape[-2]
        return torch.Size([len(self.parts), n, n])

    def add_noise(self, noise):
        noise = noise.reshape(-1)
        return LazyBatchedAdditiveKernel([p_.add_noise(noise[i if noise.numel() > 1 else 0]) for i, p_ in enumerate(self.parts)])

    def diagonal(self, *a, **k):
        return torch.stack([p_.diagonal() for p_ in self.parts])

    def log_prob_batch(self, y):
        y = y.reshape(len(self.parts), -1)
        return torch.cat([p_.log_prob_batch(y[i]) for i, p_ in enumerate(self.parts)])

    def posterior(self, y, xs):
        y = y.reshape(len(self.parts), -1)
        out = [p_.posterior(y[i], xs) for i, p_ in enumerate(self.parts)]
        return torch.cat([o[0] for o in out], 0), torch.cat([o[1] for o in out], 0)

    def evaluate(self):
        return torch.stack([p_.evaluate() for p_ in self.parts])

    to_dense = evaluate
