"""`VariationalMultitaskGPModel` (SVGP-LMC) and the gpytorch-named pieces it is built from
(reference: projected_lmc.py:659-813; `gp.mlls.VariationalELBO` at experiments.py:236).

The n-dependent arithmetic (K_ZZ, its Cholesky factor, K_ZX, the interpolation term and all their
adjoints) runs in the HIP library through `_var_engine.WhitenedInterp`; the remaining O(q m n)
contractions with the variational parameters are plain torch ops on the device.

[gpytorch-knowledge] restated semantics (whitened VariationalStrategy, LMCVariationalStrategy,
Gaussian expected_log_prob, VariationalELBO) are listed in oracle/variational.py.
"""
import math
import warnings

import numpy as np
import torch

from . import _var_engine
from . import kernels as _k
from . import means as _m
from .distributions import MultivariateNormal, MultitaskMultivariateNormal
from .models import handle_covar_, init_lmc_coefficients
from .projected import _DiagonalTaskCovariance


def variational_jitter(dtype):
    """settings.variational_cholesky_jitter defaults [gpytorch-knowledge]."""
    return 1e-4 if dtype == torch.float32 else 1e-6


class CholeskyVariationalDistribution(torch.nn.Module):
    def __init__(self, num_inducing_points, batch_shape=torch.Size(), mean_init_std=1e-3, **kwargs):
        super().__init__()
        self.num_inducing_points, self.batch_shape, self.mean_init_std = num_inducing_points, torch.Size(batch_shape), mean_init_std
        self.register_parameter("variational_mean", torch.nn.Parameter(torch.zeros(*batch_shape, num_inducing_points)))
        eye = torch.eye(num_inducing_points).repeat(*batch_shape, 1, 1)
        self.register_parameter("chol_variational_covar", torch.nn.Parameter(eye))

    def initialize_variational_distribution(self, prior_chol=None):
        """q(u) <- prior plus mean noise of std mean_init_std (first call only).  Whitened strategy: the prior is
        N(0, I) (nothing to copy); unwhitened: `prior_chol` = Cholesky factor of the prior covariance."""
        with torch.no_grad():
            self.variational_mean.add_(self.mean_init_std * torch.randn_like(self.variational_mean))
            if prior_chol is not None:
                self.chol_variational_covar.copy_(prior_chol.to(self.chol_variational_covar.dtype))


class VariationalStrategy(torch.nn.Module):
    """Whitened variational strategy; inducing points shared by the q latents."""

    def __init__(self, model, inducing_points, variational_distribution, learn_inducing_locations=True, jitter_val=None):
        super().__init__()
        object.__setattr__(self, "model", model)
        if learn_inducing_locations:
            self.register_parameter("inducing_points", torch.nn.Parameter(inducing_points.clone()))
        else:
            self.register_buffer("inducing_points", inducing_points.clone())
        self._variational_distribution = variational_distribution
        self.register_buffer("variational_params_initialized", torch.tensor(0))
        self.jitter_val = jitter_val

    def _initialized(self):
        """`variational_params_initialized` (a device buffer, as in gpytorch: part of the state dict) without a device-to-host
        sync per training step: once it has been seen set it stays set for this module object (it is only ever raised; a
        load_state_dict that lowers it comes with a fresh look because the buffer's version changes)."""
        b = self.variational_params_initialized
        seen = self.__dict__.get("_init_seen")
        if seen is not None and seen == (b._version, b.data_ptr()):
            return True
        done = bool(b.item())
        if done:
            self.__dict__["_init_seen"] = (b._version, b.data_ptr())
        return done

    def latent_moments(self, x):
        """(mean_f (q,n), var_f (q,n)) of q(f) at x and KL(q(u) || p(u)) (q,)."""
        model = self.model
        if not self._initialized():
            self._variational_distribution.initialize_variational_distribution()
            self.variational_params_initialized.fill_(1)
        kern = model.covar_module
        Z = self.inducing_points
        kind, ell, osc = kern._pieces(Z.shape[-1])
        dt = ell.dtype
        jit = self.jitter_val if self.jitter_val is not None else variational_jitter(dt)
        A = _var_engine.whitened_interp(kind, kern.select(Z), kern.select(x), ell, osc, jit)      # (q,m,n)
        vd = self._variational_distribution
        mvar = vd.variational_mean.to(dt)
        Ls = vd.chol_variational_covar.to(dt).tril()
        mean_f = (A.transpose(-1, -2) @ mvar.unsqueeze(-1)).squeeze(-1)
        Bm = _var_engine.lower_t_matmul(Ls, A)                                    # Ls^T A, triangular contraction ranges
        q = ell.shape[0]
        os_ = torch.ones(q, dtype=dt, device=ell.device) if osc is None else osc
        var_f = os_[:, None] + jit - (A * A).sum(-2) + (Bm * Bm).sum(-2)
        m = Z.shape[-2]
        logdetS = 2.0 * torch.log(torch.diagonal(Ls, dim1=-2, dim2=-1).abs()).sum(-1)
        kl = 0.5 * ((Ls * Ls).sum((-2, -1)) + (mvar * mvar).sum(-1) - m - logdetS)
        return mean_f, var_f, kl

    def kl_divergence(self):
        return self._last_kl

    def forward(self, x, **kwargs):
        mean_f, var_f, kl = self.latent_moments(x)
        self._last_kl = kl
        return MultivariateNormal(mean_f, torch.diag_embed(var_f))

    __call__ = forward


class UnwhitenedVariationalStrategy(VariationalStrategy):
    """q(u) = N(m, S) on the un-whitened inducing values; the reference selects it with inducing points = the
    training inputs when `train_ind_ratio == 1` (projected_lmc.py:724-729).  gpytorch 1.11 semantics restated
    [gpytorch-knowledge, unverified offline]:
      * prior p(u) = N(mean(Z), K_ZZ + 1e-3 I)  (`lazy_covariance_matrix.add_jitter()` default);
      * first call: q(u) <- the prior (mean + mean_init_std noise, Cholesky factor of the prior covariance);
      * x equal to the inducing points: q(f) = q(u) itself; otherwise the usual marginalisation
        mean = K_xZ Khat^-1 m,  cov = K_xx - K_xZ Khat^-1 K_Zx + K_xZ Khat^-1 S Khat^-1 K_Zx;
      * KL = KL(q(u) || p(u)) in closed form.
    The n x n factorisations run on the HIP sweep (_var_engine.GaussianKLToKernelPrior)."""

    PRIOR_JITTER = 1e-3

    def _is_inducing_points(self, x, Z):
        """x == Z without a host sync per call: same object / same memory is equal without looking; otherwise torch.equal (a
        device-to-host sync) runs once per (x, Z) version pair and is remembered (the training loop calls with the same
        tensors every step; an in-place edit of either bumps its version and brings the comparison back)."""
        if x is Z or (x.shape == Z.shape and x.dtype == Z.dtype and x.device == Z.device and x.data_ptr() == Z.data_ptr()
                      and x.stride() == Z.stride()):
            return True
        if x.shape != Z.shape:
            return False
        seen = self.__dict__.get("_xz_seen")
        if seen is not None and seen[0]() is x and seen[1] == x._version and seen[2]() is Z and seen[3] == Z._version:
            return seen[4]
        eq = bool(torch.equal(x, Z))
        import weakref
        self.__dict__["_xz_seen"] = (weakref.ref(x), x._version, weakref.ref(Z), Z._version, eq)
        return eq

    def latent_moments(self, x):
        model = self.model
        kern = model.covar_module
        Z = self.inducing_points
        kind, ell, osc = kern._pieces(Z.shape[-1])
        dt = ell.dtype
        jit = self.jitter_val if self.jitter_val is not None else self.PRIOR_JITTER
        vd = self._variational_distribution
        Zs = kern.select(Z)
        if not self._initialized():
            with torch.no_grad():
                Lp = _var_engine.prior_cholesky(kind, Zs, ell, osc, jit)
            vd.initialize_variational_distribution(prior_chol=Lp)
            self.variational_params_initialized.fill_(1)
        mvar = vd.variational_mean.to(dt)
        Ls = vd.chol_variational_covar.to(dt).tril()
        kl = _var_engine.gaussian_kl_to_kernel_prior(kind, Zs, ell, osc, mvar, Ls, jit)
        if self._is_inducing_points(x, Z):
            return mvar, (Ls * Ls).sum(-1), kl
        if torch.is_grad_enabled() and any(t.requires_grad for t in (ell, mvar, Ls)):
            raise NotImplementedError("UnwhitenedVariationalStrategy: gradients are built for x == inducing points (the "
                                      "reference's training call); evaluate other inputs under torch.no_grad()")
        mean_f, var_f = _var_engine.unwhitened_predictive(kind, Zs, kern.select(x), ell, osc, mvar, Ls, jit)
        return mean_f, var_f, kl


class LMCVariationalStrategy(torch.nn.Module):
    def __init__(self, base_variational_strategy, num_tasks, num_latents=1, latent_dim=-1, jitter_val=None):
        super().__init__()
        self.base_variational_strategy = base_variational_strategy
        self.num_tasks, self.num_latents, self.latent_dim = num_tasks, num_latents, latent_dim
        self.register_parameter("lmc_coefficients", torch.nn.Parameter(torch.randn(num_latents, num_tasks)))

    def kl_divergence(self):
        return self.base_variational_strategy.kl_divergence().sum(dim=self.latent_dim)

    def __call__(self, x, task_indices=None, prior=False, **kwargs):
        mean_f, var_f, kl = self.base_variational_strategy.latent_moments(x)
        self.base_variational_strategy._last_kl = kl
        H = self.lmc_coefficients.to(mean_f.dtype)
        mean = mean_f.transpose(-1, -2) @ H                               # (n,p)
        var = var_f.transpose(-1, -2) @ (H * H)                           # marginal task variances
        return MultitaskMultivariateNormal(mean, _DiagonalTaskCovariance(var))


class CustomLMCVariationalStrategy(LMCVariationalStrategy):
    """Adds deterministic task-level means (projected_lmc.py:659-683)."""

    def __init__(self, mean_module, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.output_mean_module = mean_module

    def __call__(self, x, task_indices=None, prior=False, **kwargs):
        dist = super().__call__(x, task_indices=None, prior=False, **kwargs)
        tasks_means = self.output_mean_module(x)                          # (p,n)
        return dist.__class__(dist.mean + tasks_means.T.to(dist.mean.dtype), dist.lazy_covariance_matrix)


class VariationalMultitaskGPModel(torch.nn.Module):
    """A variational LMC model with the reference's constructor (projected_lmc.py:690-761)."""

    def __init__(self, train_x, n_latents, n_tasks, train_ind_ratio=1.5, seed=0, init_lmc_coeffs=False, train_y=None,
                 prior_scales=None, prior_width=None, mean_type=_m.ConstantMean, kernel_type=_k.RBFKernel,
                 outputscales=False, decomp=None, distrib=CholeskyVariationalDistribution, var_strat=VariationalStrategy,
                 ker_kwargs=None, **kwargs):
        super().__init__()
        if ker_kwargs is None:
            ker_kwargs = {}
        if train_x.ndimension() == 1:
            train_x = train_x.unsqueeze(-1)
        self.dim = train_x.shape[1]
        if train_y is not None and train_y.shape[1] != n_tasks:
            n_tasks = train_y.shape[1]
            warnings.warn('Number of tasks in the training labels does not match the specified number of tasks. '
                          'Defaulting to the number of tasks in the training labels.')
        if float(train_ind_ratio) == 1.:
            warnings.warn('Caution : inducing points not learned !')
            inducing_points, learn = train_x, False
            var_strat, distrib = UnwhitenedVariationalStrategy, CholeskyVariationalDistribution
        else:
            learn = True
            from scipy.stats import qmc
            n_ind_points = int(np.floor(train_x.shape[0] / train_ind_ratio))
            sampler = qmc.LatinHypercube(d=self.dim, seed=seed)
            inducing_points = torch.as_tensor(2 * sampler.random(n=n_ind_points) - 1, dtype=train_x.dtype)
        variational_distribution = distrib(inducing_points.size(-2), batch_shape=torch.Size([n_latents]))
        strategy = var_strat(self, inducing_points, variational_distribution, learn_inducing_locations=learn)
        output_mean_module = mean_type(input_size=self.dim, batch_shape=torch.Size([n_tasks]))
        self.variational_strategy = CustomLMCVariationalStrategy(output_mean_module, strategy, num_tasks=n_tasks,
                                                                 num_latents=n_latents, latent_dim=-1)
        self.covar_module = handle_covar_(kernel_type, dim=self.dim, decomp=decomp, prior_scales=prior_scales,
                                          prior_width=prior_width, n_funcs=n_latents, ker_kwargs=ker_kwargs,
                                          outputscales=outputscales)
        self.mean_module = _m.ZeroMean(batch_shape=torch.Size([n_latents]))
        self.n_tasks, self.n_latents, self.decomp = n_tasks, n_latents, decomp
        if init_lmc_coeffs and train_y is not None:
            coeffs = init_lmc_coefficients(train_y, n_latents=n_latents)
            self.variational_strategy.lmc_coefficients = torch.nn.Parameter(coeffs.to(train_y.device))

    @property
    def base_variational_strategy(self):
        return self.variational_strategy.base_variational_strategy

    def forward(self, x):
        return MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def __call__(self, x, **kwargs):
        if x.ndimension() == 1:
            x = x.unsqueeze(-1)
        return self.variational_strategy(x, **kwargs)

    def lscales(self, unpacked=True):
        cm = self.covar_module
        base = cm.base_kernel if hasattr(cm, "base_kernel") else cm
        scales = base.lengthscale.data
        return scales if unpacked else [scales]

    def outputscale(self, unpacked=False):
        res = torch.zeros((self.n_latents, 1))
        res[:, 0] = self.covar_module.outputscale.data.squeeze()
        return res.squeeze() if unpacked else res

    def lmc_coefficients(self):
        return self.variational_strategy.lmc_coefficients.data

    def compute_latent_distrib(self, x, prior=False, **kwargs):
        return self.base_variational_strategy(x, **kwargs)


class VariationalELBO(torch.nn.Module):
    """ELBO = (1/n) sum_points E_q[log p(y|f)] - beta * KL / num_data + sum(log priors) / num_data
    [gpytorch-knowledge: _ApproximateMarginalLogLikelihood.forward, v1.11]."""

    def __init__(self, likelihood, model, num_data, beta=1.0, combine_terms=True):
        super().__init__()
        self.likelihood, self.model, self.num_data, self.beta = likelihood, model, num_data, beta

    def forward(self, variational_dist_f, target, **kwargs):
        num_batch = variational_dist_f.event_shape[0]
        log_likelihood = self.likelihood.expected_log_prob(target, variational_dist_f, **kwargs).sum(-1) / num_batch
        kl = self.model.variational_strategy.kl_divergence() / (self.num_data / self.beta)
        from .priors import named_priors
        log_prior = sum((pr.log_prob(v).sum() for _, _, pr, v in named_priors(self.model)), torch.zeros_like(kl)) / self.num_data
        return log_likelihood - kl + log_prior
