"""Sparse GP regression (Titsias 2009) kernel wrapper: `InducingPointKernel`, used by the reference
when `n_inducing_points` is given (projected_lmc.py:302-303; all real-data runs,
realdata_experiments.py:398,505) -- SURVEY.md 8f row 2.

[gpytorch-knowledge] semantics restated: in train mode the prior covariance is the Nystrom
approximation Q = K_xz K_zz^-1 K_zx (a low-rank root), the likelihood adds sigma^2 I, `log_prob`
is the exact Gaussian density under Q + sigma^2 I, and an added loss term
-1/2 sum_i (k_ii - q_ii) / sigma^2 enters the MLL through `_add_other_terms`; in eval mode the
predictive moments use Q for all blocks.

HIP path: A = L^-1 K_zx (L L^T = K_zz) is one augmented sweep of the library
(`_var_engine.WhitenedInterp`, including the adjoint w.r.t. inducing points and lengthscales); what
remains is the m x m Woodbury matrix B = I + A A^T / sigma^2: its quadratic form, log-determinant and half-solves go
through the same blocked sweep (`_dense.py`), with the closed-form adjoint -- no library factorisation on this path.
"""
import math

import torch

from . import _dense
from . import _var_engine
from .kernels import Kernel


class InducingPointKernel(Kernel):
    def __init__(self, base_kernel, inducing_points, likelihood, active_dims=None, jitter=0.0):
        super().__init__(active_dims=active_dims)
        self.base_kernel = base_kernel
        self.likelihood = likelihood
        if inducing_points.ndimension() == 1:
            inducing_points = inducing_points.unsqueeze(-1)
        self.register_parameter("inducing_points", torch.nn.Parameter(inducing_points))
        self.jitter = jitter
        self._added_loss = None

    def _pieces(self, d):
        return self.base_kernel._pieces(d)

    def select(self, x):
        return self.base_kernel.select(x)

    def forward(self, x1, x2=None, **params):
        x1 = self.select(x1)
        kind, ell, osc = self._pieces(x1.shape[-1])
        return LazySgprKernel(self, kind, x1, self.select(self.inducing_points).to(x1.dtype), ell, osc)

    def added_loss_term(self):
        t, self._added_loss = self._added_loss, None
        return t


class LazySgprKernel:
    """Un-evaluated Nystrom covariance Q(x,x) (+ noise I once the likelihood was applied)."""

    def __init__(self, owner, kind, x, Z, ell, oscale, noise=None):
        self.owner, self.kind, self.x, self.Z, self.ell, self.oscale, self.noise = owner, kind, x, Z, ell, oscale, noise
        self.x1 = x

    @property
    def shape(self):
        q, n = self.ell.shape[0], self.x.shape[-2]
        return torch.Size([q, n, n])

    def add_noise(self, noise):
        return LazySgprKernel(self.owner, self.kind, self.x, self.Z, self.ell, self.oscale,
                              noise if self.noise is None else self.noise + noise)

    def interp(self, x=None):
        return _var_engine.whitened_interp(self.kind, self.Z, self.x if x is None else x, self.ell, self.oscale,
                                           self.owner.jitter)

    def diagonal(self, *a, **k):
        A = self.interp()
        dg = (A * A).sum(-2)
        if self.noise is not None:
            dg = dg + self.noise.reshape(-1, 1)
        return dg

    def log_prob_batch(self, y):
        """log N(y_i; 0, Q_i + s_i I) for the q latents via Woodbury, y: (q, n); also records the
        SGPR added-loss term for the MLL."""
        if self.noise is None:
            raise RuntimeError("log_prob of a noise-free SGPR prior: apply the likelihood first")
        A = self.interp()                                              # (q,m,n)
        q, m, n = A.shape
        s = self.noise.reshape(q, 1, 1).to(A.dtype)
        eye = torch.eye(m, dtype=A.dtype, device=A.device)
        Bm = eye + (A @ A.transpose(-1, -2)) / s
        Ay = (A @ y.unsqueeze(-1)).squeeze(-1)                         # (q,m)
        cc, logdetB = _dense.spd_quad_logdet(Bm, Ay)                   # (A y)^T B^-1 (A y), log det B
        s1 = s.reshape(q)
        quad = ((y * y).sum(-1) - cc / s1) / s1
        logdet = n * torch.log(s1) + logdetB
        os_ = torch.ones(q, dtype=A.dtype, device=A.device) if self.oscale is None else self.oscale
        self.owner._added_loss = -0.5 * (n * os_ - (A * A).sum((-2, -1))) / s1
        return -0.5 * (quad + logdet + n * math.log(2.0 * math.pi))

    def posterior(self, y, xs, full_cov=False):
        """Predictive mean (q,ns) and variance (q,ns) -- or, with full_cov, covariance (q,ns,ns) -- of the SGPR posterior at xs:
        Q** - Q*n (Qnn + s2 I)^-1 Qn* = A*^T B^-1 A* = V*^T V* with B = I + A A^T / s2 = L_B L_B^T, V* = L_B^-1 A*."""
        A = self.interp()
        As = self.interp(xs)
        q, m, n = A.shape
        s = self.noise.reshape(q, 1, 1).to(A.dtype)
        eye = torch.eye(m, dtype=A.dtype, device=A.device)
        sol = _dense.spd_half_solve(eye + (A @ A.transpose(-1, -2)) / s, torch.cat([A @ y.unsqueeze(-1), As], -1))
        c, Vs = sol[..., :1], sol[..., 1:]                                                # L_B^-1 [A y | A*]
        mean = (Vs.transpose(-1, -2) @ c).squeeze(-1) / s.reshape(q, 1)
        if full_cov:
            return mean, Vs.transpose(-1, -2) @ Vs
        var = (Vs * Vs).sum(-2)
        return mean, var
