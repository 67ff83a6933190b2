"""Minimal Gaussian distribution types with the attribute surface the reference's drivers and
model code touch: `.mean`, `.variance`, `.stddev`, `.confidence_region()`,
`.lazy_covariance_matrix`, `.covariance_matrix`, `.event_shape`, `.batch_shape`, `.log_prob`.

`covar` may be a dense tensor, a kernels.LazyKernel (prior of a batch of GPs: log_prob goes to
the HIP engine) or one of the structured covariances below (never materialised unless asked).
"""
import torch

from .kernels import LazyKernel


class KroneckerSumCovariance:
    """sum_i C_i (x) h_i h_i^T + eps I, data-major interleaved ((n p) x (n p)) -- the task-space
    covariance ProjectedGPModel.__call__ builds at projected_lmc.py:1149-1153.  Only its diagonal is
    needed by the drivers (experiments.py:329); `evaluate()` materialises for small cases.
    C_i given either as full (q,n,n) `cov` or as diagonal (q,n) `var`."""

    def __init__(self, Ht, cov=None, var=None, eps=0.0, task_noise=None, diag=None):
        self.Ht, self.cov, self.var, self.eps, self.task_noise = Ht, cov, var, eps, task_noise
        self.diag = diag                                    # (n, p) diagonal incl. eps, if already mixed (plmc_mix_posterior)
        self.n = (cov if cov is not None else var).shape[-1]
        self.p = Ht.shape[-1]

    @property
    def shape(self):
        return torch.Size([self.n * self.p, self.n * self.p])

    def add_task_noise(self, Sigma):
        tn = Sigma if self.task_noise is None else self.task_noise + Sigma
        return KroneckerSumCovariance(self.Ht, self.cov, self.var, self.eps, tn, self.diag)

    def diagonal(self, *a, **k):
        if self.diag is not None:
            dg = self.diag
        else:
            v = self.var if self.var is not None else torch.diagonal(self.cov, dim1=-2, dim2=-1)   # (q,n)
            dg = v.T @ (self.Ht * self.Ht) + self.eps                                              # (n,p)
        if self.task_noise is not None:
            dg = dg + torch.diagonal(self.task_noise)[None, :]
        return dg.reshape(-1)

    def evaluate(self):
        if self.cov is None:
            raise RuntimeError("full task covariance requested but only latent variances were computed; "
                               "call the model with full_cov=True")
        q, n, p = self.Ht.shape[0], self.n, self.p
        B = self.Ht[:, :, None] * self.Ht[:, None, :]                                           # (q,p,p)
        C = torch.einsum("qab,qst->asbt", self.cov, B).reshape(n * p, n * p)
        C = C + self.eps * torch.eye(n * p, dtype=C.dtype, device=C.device)
        if self.task_noise is not None:
            C = C + torch.kron(torch.eye(n, dtype=C.dtype, device=C.device), self.task_noise)
        return C

    to_dense = evaluate


class MultivariateNormal:
    def __init__(self, mean, covariance_matrix, validate_args=False):
        self.loc = mean
        self._covar = covariance_matrix

    # -- shapes
    @property
    def event_shape(self):
        return self.loc.shape[-1:]

    @property
    def batch_shape(self):
        return self.loc.shape[:-1]

    @property
    def mean(self):
        return self.loc

    @property
    def lazy_covariance_matrix(self):
        return self._covar

    @property
    def covariance_matrix(self):
        c = self._covar
        return c if torch.is_tensor(c) else c.evaluate()

    @property
    def variance(self):
        c = self._covar
        if torch.is_tensor(c):
            return torch.diagonal(c, dim1=-2, dim2=-1)
        return c.diagonal().reshape(self.loc.shape)

    @property
    def stddev(self):
        return self.variance.sqrt()

    def confidence_region(self):
        s2 = self.stddev * 2.0
        return self.mean - s2, self.mean + s2

    def log_prob(self, value):
        """-1/2 (quad + logdet + n log 2pi).  Lazy kernel covariances run on the HIP engine
        (the call the reference makes at projected_lmc.py:1201)."""
        c = self._covar
        diff = value - self.loc
        if hasattr(c, "log_prob_batch"):                       # SGPR / Nystrom prior (sgpr.py)
            return c.log_prob_batch(diff.reshape(-1, diff.shape[-1])).reshape(self.batch_shape)
        if isinstance(c, LazyKernel):
            from . import _engine
            if c.noise is None:
                raise RuntimeError("log_prob of a noise-free kernel prior: apply the likelihood first")
            q = c.ell.shape[0]
            y = diff.reshape(q, -1)
            lp = _engine.exact_latent_log_prob(c.kind, c.x1, c.ell, c.oscale, c.noise.reshape(-1), y)
            return lp.reshape(self.batch_shape)
        raise NotImplementedError("log_prob is implemented for kernel priors (HIP engine) only; "
                                  "dense / predictive covariances are out of the hot-path scope")


class MultitaskMultivariateNormal(MultivariateNormal):
    """mean (n, p); covariance over the data-major interleaved vector (flat = i_point * p + i_task)
    [gpytorch-knowledge: MultitaskMultivariateNormal, interleaved=True]."""

    def __init__(self, mean, covariance_matrix, validate_args=False, interleaved=True, independent=None):
        super().__init__(mean, covariance_matrix)
        self._independent = independent      # batch MVN it was built from (from_batch_mvn)

    @classmethod
    def from_batch_mvn(cls, batch_mvn, task_dim=-1):
        """Independent tasks from a batch of q MVNs (projected_lmc.py:318-319): mean (q,n)->(n,q)."""
        return cls(batch_mvn.mean.transpose(-1, -2), batch_mvn.lazy_covariance_matrix, independent=batch_mvn)

    @property
    def event_shape(self):
        return self.loc.shape[-2:]

    @property
    def batch_shape(self):
        return self.loc.shape[:-2]

    @property
    def num_tasks(self):
        return self.loc.shape[-1]

    @property
    def variance(self):
        c = self._covar
        if self._independent is not None:
            return self._independent.variance.transpose(-1, -2)
        if torch.is_tensor(c):
            return torch.diagonal(c, dim1=-2, dim2=-1).reshape(self.loc.shape)
        return c.diagonal().reshape(self.loc.shape)

    def log_prob(self, value):
        if self._independent is not None:
            return self._independent.log_prob(value.transpose(-1, -2)).sum(-1)
        c = self._covar
        if hasattr(c, "log_prob_flat"):
            return c.log_prob_flat((value - self.loc))
        flat = MultivariateNormal(self.loc.reshape(*self.batch_shape, -1), c)
        return flat.log_prob(value.reshape(*self.batch_shape, -1))
