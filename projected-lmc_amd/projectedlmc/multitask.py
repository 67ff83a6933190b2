"""`MultitaskGPModel`: exact ICM / naive-LMC multitask GP (reference: projected_lmc.py:438-656) on
the HIP dense-LMC engine (`_lmc_engine`, csrc/lmc.hip): the (n p) x (n p) Kronecker-sum
covariance is assembled, factorised and differentiated by hand-written kernels.

Deviation from the literal reference: the SVD initialisation at :472/:476
assigns a q x p tensor where gpytorch's IndexKernel expects p x rank, which can only run when
p == q; the documented intent is implemented instead (ICM: covar_factor = coeffs^T (p x q);
LMC: latent i gets coeffs[i] as its p x 1 factor), so `lmc_coefficients()` returns the q x p SVD
loadings right after construction.
"""
import copy

import torch

from . import _lmc_engine
from . import kernels as _k
from . import means as _m
from .distributions import MultitaskMultivariateNormal
from .models import ExactGPModel, init_lmc_coefficients
from .projected import _DiagonalTaskCovariance


class MultitaskGPModel(ExactGPModel):
    def __init__(self, train_x, train_y, likelihood, n_tasks, n_latents, model_type='ICM', init_lmc_coeffs=True,
                 fix_diagonal=False, **kwargs):
        super().__init__(train_x, train_y, likelihood, n_tasks=1, outputscales=False, **kwargs)
        self.mean_module = _m.MultitaskMean(self.mean_module, num_tasks=n_tasks)
        base = self.covar_module
        if model_type == 'ICM':
            self.covar_module = _k.MultitaskKernel(base, num_tasks=n_tasks, rank=n_latents)
        elif model_type == 'LMC':
            self.covar_module = _k.LCMKernel(base_kernels=[copy.deepcopy(base) for _ in range(n_latents)],
                                             num_tasks=n_tasks, rank=1)
        else:
            raise ValueError('Wrong specified model type, should be ICM or LMC')

        if init_lmc_coeffs:
            coeffs = init_lmc_coefficients(train_y, n_latents)                      # q x p
            if model_type == 'ICM':
                self.covar_module.task_covar_module.covar_factor = torch.nn.Parameter(coeffs.T.contiguous())
            else:
                for i in range(n_latents):
                    self.covar_module.covar_module_list[i].task_covar_module.covar_factor = torch.nn.Parameter(
                        coeffs[i].unsqueeze(-1).contiguous())
        if fix_diagonal:
            mods = [self.covar_module] if model_type == 'ICM' else list(self.covar_module.covar_module_list)
            for mod in mods:
                mod.task_covar_module.raw_var = torch.nn.Parameter(
                    -10 * torch.ones(n_tasks, device=train_y.device, dtype=train_y.dtype), requires_grad=False)
        self.n_tasks, self.n_latents, self.model_type = n_tasks, n_latents, model_type

    # ------------------------------------------------------------------ inspection helpers
    def lmc_coefficients(self):
        """(n_latents x n_tasks) LMC / ICM coefficients (projected_lmc.py:493-505)."""
        if self.model_type == 'LMC':
            res = torch.zeros((self.n_latents, self.n_tasks))
            for i in range(self.n_latents):
                res[i] = self.covar_module.covar_module_list[i].task_covar_module.covar_factor.data.squeeze()
            return res
        return self.covar_module.task_covar_module.covar_factor.data.squeeze().T

    def _data_kernels(self):
        if self.model_type == 'LMC':
            return [m.data_covar_module for m in self.covar_module.covar_module_list]
        return [self.covar_module.data_covar_module]

    def lscales(self, unpacked=True):
        ks = self._data_kernels()
        base = [k.base_kernel if hasattr(k, "base_kernel") else k for k in ks]
        if self.model_type == 'ICM':
            scales = base[0].lengthscale.data.squeeze().repeat(self.n_latents, 1)
        else:
            scales = torch.stack([b.lengthscale.data.squeeze().reshape(-1) for b in base])
        return scales if unpacked else [scales]

    def outputscale(self, unpacked=False):
        res = torch.zeros((self.n_latents, 1))
        ks = self._data_kernels()
        if self.model_type == 'LMC':
            for i, k in enumerate(ks):
                res[i, 0] = k.outputscale.data.squeeze()
        else:
            res[:, 0] = ks[0].outputscale.data.squeeze()
        return res.squeeze() if unpacked else res

    # ------------------------------------------------------------------------------ forward
    def forward(self, x):
        mean_x = self.mean_module(x)
        covar_x = self.covar_module(x)
        return MultitaskMultivariateNormal(mean_x, covar_x)

    def _posterior(self, x, **kwargs):
        """Eval mode: exact task posterior from one augmented factorisation of the dense system."""
        tx = self.train_inputs[0]
        lazy = self.covar_module(tx)
        Sigma = self.likelihood.task_noise_matrix(lazy.B.dtype)
        resid = (self.train_targets - self.mean_module(tx)).reshape(-1)
        sel = self._data_kernels()[0].select
        mean, var = _lmc_engine.lmc_posterior(lazy.kind, lazy.x, lazy.ell.detach(),
                                              None if lazy.oscale is None else lazy.oscale.detach(),
                                              lazy.B.detach(), Sigma.detach(), resid.detach(), sel(x))
        return MultitaskMultivariateNormal(mean + self.mean_module(x), _DiagonalTaskCovariance(var))

    def compute_loo(self, output=None):
        """Leave-one-out variances and residuals of the dense multitask system, shaped like train_y (n x p)
        (projected_lmc.py:642-656: K = likelihood(output), sigma2 = 1 / diag(K^-1), y - mu_loo = K^-1 (y - m) sigma2).
        `output` is accepted for signature compatibility; the covariance is rebuilt from the current parameters."""
        tx = self.train_inputs[0]
        lazy = self.covar_module(tx)
        Sigma = self.likelihood.task_noise_matrix(lazy.B.dtype)
        resid = (self.train_targets - self.mean_module(tx)).reshape(-1)
        with torch.no_grad():
            s2, r = _lmc_engine.lmc_loo(lazy.kind, lazy.x, lazy.ell.detach(), None if lazy.oscale is None else lazy.oscale.detach(),
                                        lazy.B.detach(), Sigma.detach(), resid.detach())
        return s2.reshape(self.train_targets.shape), r.reshape(self.train_targets.shape)

    def compute_var(self, x):
        """Predictive variance incl. likelihood noise, clamped at 1e-6 (projected_lmc.py:591-640;
        the reference evaluates it through a Kronecker eigen-decomposition, ICM only)."""
        if self.model_type != 'ICM':
            raise ValueError('This method is only available for ICM models')
        was = self.training
        self.eval()
        with torch.no_grad():
            pred = self.likelihood(self(x))
            var = pred.variance
        self.train(was)
        return torch.clamp(var, min=1e-6)
