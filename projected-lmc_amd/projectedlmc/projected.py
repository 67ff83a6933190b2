"""The Projected LMC model and its loss (reference: projected_lmc.py:819-1241), on the HIP engine.

The p x q projection algebra (QR of the mixing matrix, T = H^+, noise blocks) is tiny and stays in
torch on the device, exactly where the reference keeps it; everything that touches an n x n
object -- the q latent exact-GP log-likelihoods, their gradients and the latent posteriors -- runs
in libplmc_hip.so.  With `latent_shard=(rank, world)` the q independent latent GPs are split
across ranks and `ProjectedLMCmll` returns this rank's share of the loss.
"""
import math
import os
import warnings

import torch
from torch.nn.utils import parametrize

from . import _engine
from . import settings
from . import _qr
from . import kernels as _k
from .kernels import LazyKernel
from . import means as _m
from .constraints import GreaterThan
from .distributions import MultivariateNormal, MultitaskMultivariateNormal, KroneckerSumCovariance
from .likelihoods import GaussianLikelihood, MultitaskGaussianLikelihood, _GaussianLikelihoodBase
from .mlls import ExactMarginalLogLikelihood
from .models import (ExactGP, ExactGPModel, init_lmc_coefficients, ScalarParam, PositiveDiagonalParam,
                     UpperTriangularParam, LowerTriangularParam)


import weakref

# Step-local caches (QR of the mixing matrix, Q^T Y^T) live OUTSIDE the modules: they hold non-leaf tensors, and a
# parametrized module is deep-copied through its __dict__ (torch.nn.utils.parametrize), __getstate__ or not.
_STEP_CACHE = weakref.WeakKeyDictionary()


def _cache_get(mod, name):
    return _STEP_CACHE.get(mod, {}).get(name)


def _cache_set(mod, name, value):
    if value is None:
        _STEP_CACHE.get(mod, {}).pop(name, None)
    else:
        _STEP_CACHE.setdefault(mod, {})[name] = value


class LMCMixingMatrix(torch.nn.Module):
    """Mixing matrix H = Q R (p x q), stored either in bulk (free matrix `H`, re-factored by QR at
    every call) or as separately parametrised `Q_plus`, `R` (projected_lmc.py:819-890)."""

    def __init__(self, Q_plus, R, bulk=True):
        super().__init__()
        p, c = Q_plus.shape
        if c == p:
            self.mode = 'Q_plus'
        elif c == R.shape[0]:
            self.mode = 'Q'
        else:
            raise ValueError('Wrong dimensions for Q_plus : should be n_tasks x n_tasks or n_tasks x n_latents')
        self.n_latents, self.n_tasks = R.shape[0], p
        self._size = torch.Size([self.n_latents, self.n_tasks])
        self.bulk = bulk
        if bulk:
            if self.mode == 'Q_plus':
                R_pad = torch.eye(p, dtype=R.dtype, device=R.device)
                R_pad[:self.n_latents, :self.n_latents] = R
                H = Q_plus @ R_pad
            else:
                H = Q_plus @ R
            self.register_parameter("H", torch.nn.Parameter(H))
        else:
            self.register_parameter("Q_plus", torch.nn.Parameter(Q_plus))
            self.register_parameter("R", torch.nn.Parameter(R))

    def Q(self):
        return self.Q_plus[:, :self.n_latents] if self.mode == 'Q_plus' else self.Q_plus

    def Q_orth(self):
        return self.Q_plus[:, self.n_latents:]

    def QR(self, store=False, reuse=False):
        """(Q, R, Q_orth).  Bulk mode re-factors H at every call in the reference (:864-875), and a training step calls
        it twice on the same H (projection :1015, MLL terms :1208); the factorisation is one device launch plus its
        backward, so ONE result is shared inside a training step and nowhere else: `project_data` stores it
        (store=True, only while training with grad enabled), `ProjectedLMCmll.forward` picks it up (reuse=True) and
        drops it.  Every other caller (eval-mode posterior, full_likelihood, projection_matrix) factors afresh, as the
        reference does -- a cache there would survive in-place edits through H.data (no version bump), hold non-leaf
        tensors in the module (deepcopy fails) and hand an already back-propagated graph to the next step."""
        q = self.n_latents
        if self.bulk:
            key = (self.H._version, self.H.data_ptr(), self.H.dtype, self.H.device)
            cached = _cache_get(self, "qr") if reuse else None
            if cached is not None and cached[0] == key:
                _, Qf, Rf = cached
            else:
                Qf, Rf = _qr.qr(self.H)
                if store and self.training and torch.is_grad_enabled():
                    _cache_set(self, "qr", (key, Qf, Rf))
            if self.mode == 'Q_plus':
                return Qf[:, :q], Rf[:q, :q], Qf[:, q:]
            return Qf, Rf, None
        return self.Q(), self.R, self.Q_orth()

    def drop_qr_cache(self):
        _cache_set(self, "qr", None)

    def forward(self):
        if self.bulk:
            return self.H.T if self.mode == 'Q' else self.H[:, :self.n_latents].T
        return (self.Q() @ self.R).T

    def size(self, int=None):
        return self._size[int] if int else self._size


class ProjectedGPModel(ExactGPModel):
    """The projected LMC of the reference article (projected_lmc.py:893-1155)."""

    def __init__(self, train_x, train_y, n_tasks, n_latents, proj_likelihood=None, init_lmc_coeffs=False, BDN=True,
                 diagonal_B=False, scalar_B=False, diagonal_R=False, mean_type=_m.ConstantMean,
                 ortho_param='matrix_exp', bulk=True, noise_thresh=-9., noise_init=1e-2, outputscales=False,
                 eps=1e-3, latent_shard=None, **kwargs):
        if proj_likelihood is None or proj_likelihood.noise.shape[0] != n_latents:
            warnings.warn("In projected GP model the dimension of the likelihood is the number of latent processes. "
                          "Provided likelihood was the wrong shape or None, so it was replaced by a fresh one")
            proj_likelihood = GaussianLikelihood(batch_shape=torch.Size([n_latents]),
                                                 noise_constraint=GreaterThan(math.exp(noise_thresh)))
        super().__init__(train_x, torch.zeros_like(train_y), proj_likelihood, n_tasks=n_latents,
                         mean_type=_m.ZeroMean, outputscales=outputscales, **kwargs)
        self.register_buffer('train_y', train_y)
        if mean_type is not _m.ZeroMean:
            raise ValueError('Projected GP model does not support non-zero output-wise means for now !')

        n_data, n_tasks = train_y.shape
        dt, dev = train_y.dtype, train_y.device
        if init_lmc_coeffs:
            if scalar_B and BDN:
                Q_plus, R = init_lmc_coefficients(train_y, n_latents=n_latents, QR_form=True)
            else:
                Q_plus, R_padded = init_lmc_coefficients(train_y, n_latents=n_tasks, QR_form=True)
                R = R_padded[:n_latents]
        else:
            fake_coeffs = torch.randn(n_tasks, n_latents)
            Q_plus, R_padded, _ = torch.linalg.svd(fake_coeffs)
            Q_plus, R = Q_plus.to(dt).to(dev), R_padded[:n_latents].to(dt).to(dev)
            if scalar_B and BDN:
                Q_plus = Q_plus[:, :n_latents]
        R = torch.diag_embed(R) / math.sqrt(n_data - 1)
        lmc = LMCMixingMatrix(Q_plus, R, bulk=bulk)
        if not bulk:
            lmc = torch.nn.utils.parametrizations.orthogonal(lmc, name="Q_plus", orthogonal_map=ortho_param,
                                                             use_trivialization=(ortho_param != 'householder'))
            parametrize.register_parametrization(lmc, "R", PositiveDiagonalParam() if diagonal_R
                                                 else UpperTriangularParam())
        self.lmc_coefficients = lmc

        pq = n_tasks - n_latents
        log_init = math.log(noise_init)
        if scalar_B:
            diagonal_B = True
            self.register_parameter("log_B_tilde", torch.nn.Parameter(log_init * torch.ones(pq, dtype=dt, device=dev)))
            parametrize.register_parametrization(self, "log_B_tilde", ScalarParam(bounds=(noise_thresh, -noise_thresh)))
            if BDN:
                self.register_buffer('Y_squared_norm', (train_y ** 2).sum())
        elif diagonal_B:
            self.register_parameter("log_B_tilde", torch.nn.Parameter(log_init * torch.ones(pq, dtype=dt, device=dev)))
            self.log_B_tilde_constraint = GreaterThan(noise_thresh)      # registered, never applied (:981)
        else:
            self.register_parameter("B_tilde_inv_chol", torch.nn.Parameter(
                torch.diag_embed(-log_init * torch.ones(pq, dtype=dt, device=dev))))
            parametrize.register_parametrization(self, "B_tilde_inv_chol",
                                                 LowerTriangularParam(bounds=(noise_thresh, -noise_thresh)))
        self.diagonal_B, self.scalar_B = diagonal_B, scalar_B
        if not BDN:
            self.register_parameter("M", torch.nn.Parameter(torch.zeros((n_latents, pq), dtype=dt, device=dev)))
        self.n_tasks, self.n_latents = n_tasks, n_latents
        self.latent_dim = -1
        self.eps = eps
        self.set_latent_shard(latent_shard)

    # ------------------------------------------------------------------ latent sharding (multi-GPU)
    def set_latent_shard(self, shard):
        """shard = (rank, world) -> this process owns latents rank, rank+world, ...  (None = all)."""
        if shard is None:
            self.latent_ids = None
        else:
            rank, world = shard
            if not (0 <= rank < world):
                raise ValueError("latent_shard = (rank, world) needs 0 <= rank < world, got %r" % (shard,))
            if world > self.n_latents:
                raise ValueError("latent_shard: %d ranks for %d latent processes -- ranks beyond the number of latents "
                                 "would own nothing and block the all-reduce; shard over at most n_latents ranks"
                                 % (world, self.n_latents))
            self.latent_ids = list(range(rank, self.n_latents, world))
        self.latent_shard = shard

    # ------------------------------------------------------------------------------ small algebra
    def projected_noise(self):
        return self.likelihood.noise.squeeze(-1)

    def projection_matrix(self):
        """T (p x q) with Y T = projected data (projected_lmc.py:1003-1012)."""
        Q, R, Q_orth = self.lmc_coefficients.QR()
        H_pinv = torch.linalg.solve_triangular(R.T, Q, upper=False, left=False)
        if hasattr(self, "M"):
            return H_pinv + Q_orth @ self.M.T * self.projected_noise()[None, :]
        return H_pinv

    def project_data(self, data):
        """(q x n) projected observations (projected_lmc.py:1014-1021)."""
        Q, R, Q_orth = self.lmc_coefficients.QR(store=True)
        QtY = Q.T @ data.T                                               # q x n
        # ProjectedLMCmll needs |Y Q|_F^2 of the same Y and Q (scalar-B discarded-noise term, :1215): keep the product
        _cache_set(self, "QtY", (Q, data, QtY) if (self.training and torch.is_grad_enabled()) else None)
        out = torch.linalg.solve_triangular(R, QtY, upper=True)
        if hasattr(self, "M"):
            out = out + self.projected_noise()[:, None] * self.M @ Q_orth.T @ data.T
        return out

    def _B_tilde_root(self):
        pq = self.n_tasks - self.n_latents
        if self.diagonal_B:
            return torch.diag_embed(torch.exp(self.log_B_tilde / 2))
        eye = torch.eye(pq, dtype=self.B_tilde_inv_chol.dtype, device=self.B_tilde_inv_chol.device)
        return torch.linalg.solve_triangular(self.B_tilde_inv_chol, eye, upper=False).T

    def full_noise_covariance(self):
        """Task-space noise Sigma (p x p) (projected_lmc.py:1026-1060)."""
        with parametrize.cached():
            return self._full_noise_covariance()

    def _full_noise_covariance(self):
        Q, R, Q_orth = self.lmc_coefficients.QR()
        QRm = Q @ R
        sp = self.projected_noise()
        p = self.n_tasks
        if hasattr(self, "M"):
            Bt = self._B_tilde_root()
            Bt = Bt @ Bt.T
            B_term = Q_orth @ Bt @ Q_orth.T
            M_term = -QRm @ (sp[:, None] * self.M) @ Bt @ Q_orth.T
            D_rot = torch.diag_embed(sp) + sp[:, None] * self.M @ Bt @ self.M.T * sp[None, :]
            return QRm @ D_rot @ QRm.T + M_term + M_term.T + B_term
        if self.scalar_B:
            if self.log_B_tilde.numel() > 0:
                eye = torch.eye(p, dtype=QRm.dtype, device=QRm.device)
                B_term = torch.exp(self.log_B_tilde[0]) * (eye - Q @ Q.T)
            else:
                B_term = 0.
        else:
            Br = Q_orth @ self._B_tilde_root()
            B_term = Br @ Br.T
        Droot = QRm * torch.sqrt(sp)[None, :]
        return Droot @ Droot.T + B_term

    def full_likelihood(self):
        """MultitaskGaussianLikelihood(rank=p, no global noise) whose factor is the jittered Cholesky
        of Sigma (projected_lmc.py:1023-1074)."""
        res = MultitaskGaussianLikelihood(num_tasks=self.n_tasks, rank=self.n_tasks, has_global_noise=False)
        Sigma = self.full_noise_covariance()
        res = res.to(device=Sigma.device, dtype=Sigma.dtype)
        with torch.no_grad():
            eps = 1e-6
            eye = torch.eye(self.n_tasks, dtype=Sigma.dtype, device=Sigma.device)
            while eps < self.eps:
                L, info = torch.linalg.cholesky_ex(Sigma + eps * eye)
                if int(info) == 0:
                    res.task_noise_covar_factor.data = L
                    break
                eps *= 10
                warnings.warn("Cholesky of the full noise covariance failed. Trying again with jitter {0} ...".format(eps))
        return res

    def B_tilde(self):
        if self.diagonal_B:
            return torch.diag_embed(torch.exp(self.log_B_tilde))
        r = self._B_tilde_root()
        return r @ r.T

    # ------------------------------------------------------------------------------------ forward
    def forward(self, x):
        """Prior of the latent processes only (projected_lmc.py:1088-1091)."""
        return MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def _latent_posterior(self, x, full_cov=False):
        with parametrize.cached():
            return self._latent_posterior_body(x, full_cov)

    def _latent_posterior_body(self, x, full_cov=False):
        tx = self.train_inputs[0]
        lazy = self.covar_module(tx)
        ytil = self.project_data(self.train_y).detach()
        if hasattr(lazy, "log_prob_batch"):                    # SGPR latents (n_inducing_points)
            with torch.no_grad():
                noisy = lazy.add_noise(self.projected_noise().detach().to(lazy.ell.dtype))
                return noisy.posterior(ytil, self.covar_module.select(x), full_cov=full_cov)
        ids = self.latent_ids
        ell, osc, noise = lazy.ell.detach(), lazy.oscale, self.projected_noise().detach().to(lazy.ell.dtype)
        osc = None if osc is None else osc.detach()
        if ids is not None:
            ell, noise, ytil = ell[ids], noise[ids], ytil[ids]
            osc = None if osc is None else osc[ids]
        return _engine.exact_posterior(lazy.kind, lazy.x1, ell, osc, noise, ytil, self.covar_module.select(x),
                                       full_cov=full_cov, cache=self._prediction_cache(),
                                       key=_engine.model_state_key(self, tx, self.train_y) + (tuple(ids) if ids is not None else ()))

    def compute_loo(self, output=None):
        """LOO moments of the q latent GPs on the projected data, (n x q) each
        (projected_lmc.py:1108-1119)."""
        tx = self.train_inputs[0]
        lazy = self.covar_module(tx)
        with torch.no_grad():
            ytil = self.project_data(self.train_y)
            s2, r = _engine.exact_loo(lazy.kind, lazy.x1, lazy.ell, lazy.oscale,
                                      self.projected_noise().to(lazy.ell.dtype), ytil)
        return s2.T, r.T

    def compute_latent_distrib(self, x, full_cov=True, **kwargs):
        """Posterior of the latent processes at x, mean (q x n) (projected_lmc.py:1093-1106)."""
        if self.training:
            return self.forward(x)
        mean, v = self._latent_posterior(x, full_cov=full_cov)
        return MultivariateNormal(mean, v if full_cov else torch.diag_embed(v))

    def _posterior(self, x, full_cov=False, **kwargs):
        """Task-space posterior: mean = latent_mean^T H^T, covariance = sum_i S_i (x) h_i h_i^T + eps I
        (projected_lmc.py:1133-1155).  By default only the latent variances are formed (what
        `.variance` / `.confidence_region()` need); full_cov=True keeps the n* x n* latent blocks.
        With latent sharding the partial sums are all-reduced (the one cross-latent reduction)."""
        mean_lat, v = self._latent_posterior(x, full_cov=full_cov)
        Ht = self.lmc_coefficients().detach()
        v_diag = v if not full_cov else torch.diagonal(v, dim1=-2, dim2=-1)
        if self.latent_ids is not None:
            # this rank's latents: partial sums of mean and variance (plmc_mix_posterior), then the one all-reduce
            mean, var = _engine.mix_posterior(mean_lat, v_diag, Ht[self.latent_ids], 0.0)
            from . import parallel
            buf = torch.stack([mean, var])
            parallel.all_reduce_sum(buf)
            mean, var = buf[0], buf[1] + self.eps
            return MultitaskMultivariateNormal(mean, _DiagonalTaskCovariance(var))
        mean, var = _engine.mix_posterior(mean_lat, v_diag, Ht, self.eps)
        cov = KroneckerSumCovariance(Ht, cov=v if full_cov else None, var=None if full_cov else v, eps=self.eps, diag=var)
        return MultitaskMultivariateNormal(mean, cov)


class _DiagonalTaskCovariance:
    """Marginal task variances only (sharded prediction path)."""

    def __init__(self, var, task_noise=None):
        self.var, self.task_noise = var, task_noise

    def add_task_noise(self, Sigma):
        return _DiagonalTaskCovariance(self.var, Sigma if self.task_noise is None else self.task_noise + Sigma)

    def diagonal(self, *a, **k):
        v = self.var
        if self.task_noise is not None:
            v = v + torch.diagonal(self.task_noise)[None, :]
        return v.reshape(-1)

    def evaluate(self):
        raise RuntimeError("only marginal variances are available on the sharded prediction path")


class _LateCheck:
    """State of one training step's late pivot check (settings.late_pivot_check)."""

    def __init__(self, mll, dc, args):
        self.mll, self.dc, self.args = mll, dc, args
        self.stream = torch.cuda.current_stream(args[1].device)       # the redo runs where the forward pass ran (the callback may
        self.done = False                                             # come from an autograd worker thread with its own current stream)
        self.result = None
        self.snapshot = None

    def params(self):
        return [p for p in self.mll.parameters() if p.requires_grad]

    def begin_backward(self, gout):
        # what the parameters' .grad held before this backward pass: restored if the pass has to be redone (a clone only where
        # gradients are being accumulated across calls; zero_grad() leaves None)
        if self.snapshot is None and not self.done:
            self.snapshot = [(p, None if p.grad is None else p.grad.clone()) for p in self.params()]
            self.gout = gout
            torch.autograd.Variable._execution_engine.queue_callback(self.end_backward)

    def end_backward(self):
        if self.done:
            return
        self.done = True
        snapshot, self.snapshot = self.snapshot, None
        mll, args, self.mll, self.args = self.mll, self.args, None, None
        if not self.dc.failed():                                  # waits for the copy behind the sweep, not for the GPU
            return
        warnings.warn("non-positive-definite matrix found behind the backward pass (settings.late_pivot_check): forward and backward "
                      "are redone with jitter; the tensor ProjectedLMCmll returned is overwritten with the jittered value, tensors "
                      "computed from it before backward() keep the failed pass's value", RuntimeWarning)
        for p, g in snapshot:
            p.grad = g
        with torch.enable_grad(), torch.cuda.stream(self.stream):
            # the graph behind the prior (the constrained hyper-parameters) went with the backward pass that just ended: build it
            # again -- in training mode the prior is the model at its training inputs
            args = (mll.model(*mll.model.train_inputs),) + tuple(args[1:])
            res = mll._jitter_ladder(args, self.dc)
            res.backward(self.gout)
        out = self.result() if self.result is not None else None
        if out is not None:
            with torch.no_grad():
                out.copy_(res.detach())


class _LatePivotCheck(torch.autograd.Function):
    """Identity on the loss; its backward (the first node of the backward pass) arms the end-of-backward check."""

    @staticmethod
    def forward(ctx, loss, late):
        ctx.late = late
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, g):
        ctx.late.begin_backward(g)
        return g, None


class ProjectedLMCmll(ExactMarginalLogLikelihood):
    """Loss of the ProjectedGPModel (projected_lmc.py:1158-1241): sum of the q latent exact-GP
    log-likelihoods of the projected data (HIP engine) / n, plus the projection terms."""

    def __init__(self, latent_likelihood, model):
        if not isinstance(latent_likelihood, _GaussianLikelihoodBase):
            raise RuntimeError("Likelihood must be Gaussian for exact inference")
        super().__init__(latent_likelihood, model)
        self.previous_lat = None

    def forward(self, latent_function_dist, target, inputs=None, *params):
        """The pivot check of the latent factorisation is looked at AFTER the whole forward pass has been queued (the
        host would otherwise sit out the sweep and queue the projection terms behind it); on a non-PD pivot the
        forward pass is redone with jitter -- the ladder of gpytorch's psd_safe_cholesky [gpytorch-knowledge], same
        warnings, same final error (reference call site: experiments.py:265, cholesky_max_tries)."""
        if not isinstance(latent_function_dist, MultivariateNormal):
            raise RuntimeError("ExactMarginalLogLikelihood can only operate on Gaussian random variables")
        if not (settings.check_cholesky.on() and target.is_cuda) or os.environ.get("PLMC_DEFER_CHECK", "1") == "0":
            return self._forward_once(latent_function_dist, target, inputs, *params)
        self._settle_late_check()
        args = (latent_function_dist, target, inputs) + tuple(params)
        with _engine.deferred_pivot_checks(0.0) as dc:
            res = self._forward_once(*args)
        if (settings.late_pivot_check.on() and os.environ.get("PLMC_LATE_CHECK", "1") != "0" and torch.is_grad_enabled()
                and res.requires_grad and self.model.training):
            # training step: the check moves behind the backward pass (settings.late_pivot_check)
            late = _LateCheck(self, dc, args)
            res = _LatePivotCheck.apply(res, late)
            late.result = weakref.ref(res)
            self._late = late
            return res
        if not dc.failed():
            return res
        return self._jitter_ladder(args, dc)

    def _jitter_ladder(self, args, dc):
        """The forward pass again with jitter 1e-6 (fp32) x 10^i -- gpytorch's psd_safe_cholesky [gpytorch-knowledge]."""
        base, tries = settings.cholesky_jitter.value(args[1].dtype), settings.cholesky_max_tries.value()
        jit = 0.0
        for i in range(tries):
            jit = base * (10 ** i)
            warnings.warn("A not p.d., added jitter of %.1e to the diagonal" % jit, RuntimeWarning)
            with _engine.deferred_pivot_checks(jit) as dc:
                res = self._forward_once(*args)
            if not dc.failed():
                return res
        raise RuntimeError("Matrix not positive definite after repeatedly adding jitter up to %.1e "
                           "(first failing pivot per latent: %s)" % (jit, dc.first_bad))

    def _settle_late_check(self):
        """A late check whose backward pass never ran (a loss evaluated with gradients on and dropped) is looked at now: by the
        next forward call its copy has long landed."""
        late, self._late = getattr(self, "_late", None), None
        if late is not None and not late.done:
            late.done = True
            if late.dc.failed():
                warnings.warn("the previous loss evaluation met a non-positive-definite matrix and was never back-propagated: "
                              "its value was computed without jitter (not finite)", RuntimeWarning)

    def _forward_once(self, latent_function_dist, target, inputs=None, *params):
        # every parametrised tensor (log_B_tilde / B_tilde_inv_chol; with bulk=False the orthogonal Q_plus -- a matrix exponential --
        # and R) is evaluated ONCE per forward pass instead of at every attribute access (same values, one autograd node each)
        with parametrize.cached():
            return self._forward_body(latent_function_dist, target, inputs, *params)

    def _forward_body(self, latent_function_dist, target, inputs=None, *params):
        model = self.model
        num_data = latent_function_dist.event_shape.numel()
        # (reference order :1200-1201 is projection, then likelihood; the two are independent.  The hyper-parameter
        # gradient node is created before the projection graph on purpose -- see _engine.prepare_hyper_grad.)
        latent_output = self.likelihood(latent_function_dist, *params)
        ids = model.latent_ids
        c = latent_output.lazy_covariance_matrix
        exact = isinstance(c, LazyKernel) and c.noise is not None
        if exact:
            sel = (lambda t: t) if ids is None else (lambda t: t[ids])   # this rank's latents only
            ell_s, nz_s = sel(c.ell), sel(c.noise.reshape(-1))
            osc = None if c.oscale is None else sel(c.oscale)
            hyper = _engine.prepare_hyper_grad(ell_s, osc, nz_s)
        proj_target = model.project_data(target)                         # q x n
        if exact:
            # the latent means are zero (ZeroMean is enforced at construction): nothing to subtract
            diff = proj_target if isinstance(model.mean_module, _m.ZeroMean) else proj_target - latent_output.loc
            latent_res = _engine.exact_latent_log_prob(c.kind, c.x1, ell_s, osc, nz_s, sel(diff), hyper=hyper)
        else:
            latent_res = latent_output.log_prob(proj_target)
        latent_res = self._add_other_terms(latent_res, params).sum() / num_data

        p, q = model.n_tasks, model.n_latents
        self.proj_term_list = [0] * 3
        Q, R, Q_orth = model.lmc_coefficients.QR(reuse=True)
        model.lmc_coefficients.drop_qr_cache()              # last use in a training step: do not outlive the graph
        if not hasattr(model, 'M') and model.scalar_B:
            if model.log_B_tilde.numel() > 0:
                lb = model.log_B_tilde
                root_diag = lb / 2
                cached = _cache_get(model, "QtY")                        # Q^T Y^T of project_data, same Q and Y
                YQ = cached[2] if (cached is not None and cached[0] is Q and cached[1] is target) else target @ Q
                _cache_set(model, "QtY", None)
                self.proj_term_list[1] = -0.5 * torch.exp(-lb[0]) * (model.Y_squared_norm - YQ.pow(2).sum()) / num_data
            else:
                self.proj_term_list[1] = 0.
                root_diag = torch.zeros(1, dtype=target.dtype, device=target.device)
        else:
            rot = target @ Q_orth                                        # n x (p-q)
            if model.diagonal_B:
                root_diag = model.log_B_tilde / 2
                # the reference forms the n x n matrix rot B^-1 rot^T and takes its trace (:1224,:1230);
                # the same number is the weighted squared Frobenius norm, O(n (p-q))
                self.proj_term_list[1] = -0.5 * (rot.pow(2) * torch.exp(-model.log_B_tilde)[None, :]).sum() / num_data
            else:
                Bc = model.B_tilde_inv_chol
                root_diag = -torch.log(torch.diagonal(Bc))
                self.proj_term_list[1] = -0.5 * (rot @ Bc).pow(2).sum() / num_data
        self.proj_term_list[0] = -0.5 * 2 * torch.sum(root_diag)
        if model.lmc_coefficients.bulk:
            self.proj_term_list[2] = -0.5 * torch.log(torch.diagonal(R) ** 2).sum()
        else:
            self.proj_term_list[2] = -0.5 * 2 * torch.diagonal(model.lmc_coefficients.parametrizations.R.original).sum()
        projection_term = sum(self.proj_term_list) - 0.5 * (p - q) * math.log(2 * math.pi)
        if ids is not None and model.latent_shard[0] != 0:
            # replicated terms are counted once (on rank 0) so that the all-reduced sum is the loss
            projection_term = projection_term * 0.0
        return latent_res + projection_term
