"""The few numerical settings the reference's drivers touch, as context managers with the
gpytorch names (experiments.py:265,312 `cholesky_max_tries`; README.md:54 `cholesky_jitter`).
[gpytorch-knowledge] defaults: cholesky_max_tries = 3, cholesky_jitter = 1e-6 (fp32) / 1e-8 (fp64).
"""
import torch


class _Value:
    _default = None
    _stack = None

    def __init__(self, value):
        self._value = value

    def __enter__(self):
        type(self)._stack.append(self._value)
        return self

    def __exit__(self, *exc):
        type(self)._stack.pop()
        return False


class cholesky_max_tries(_Value):
    _stack = [3]

    @classmethod
    def value(cls):
        return cls._stack[-1]


class cholesky_jitter(_Value):
    """cholesky_jitter(float_value=None, double_value=None) or cholesky_jitter(x) for both."""
    _stack = [(1e-6, 1e-8)]

    def __init__(self, float_value=None, double_value=None):
        f, d = type(self)._stack[-1]
        if float_value is not None and double_value is None:
            double_value = float_value
        super().__init__((float_value if float_value is not None else f, double_value if double_value is not None else d))

    @classmethod
    def value(cls, dtype):
        f, d = cls._stack[-1]
        return f if dtype == torch.float32 else d


class check_cholesky(_Value):
    """check_cholesky(False) skips the host read-back of the factorization status (one device
    sync per step); a non-PD matrix then surfaces as a NaN loss instead of a jitter retry."""
    _stack = [True]

    @classmethod
    def on(cls):
        return cls._stack[-1]


class late_pivot_check(_Value):
    """late_pivot_check(True | False): in a training step (ProjectedLMCmll.forward with gradients on) the pivot check of the
    latent factorisation is looked at when the BACKWARD pass has been queued, not at the end of the forward pass: the host
    queues the projection terms, the backward pass and (after the check) the optimiser step while the sweep runs, instead of
    sitting the sweep out first.  Nothing is skipped: on a non-PD pivot the gradients this backward pass accumulated are taken
    back and the jitter ladder (gpytorch's psd_safe_cholesky, same warnings, same final error) redoes forward and backward
    before `backward()` returns; the tensor ProjectedLMCmll returned is overwritten with the jittered value (a tensor computed from
    it BEFORE backward(), e.g. `loss = -mll(...)`, keeps the failed pass's value for that one step -- a warning says so; gradients
    and parameters are those of the jittered pass).  Assumes the parameters' gradients of this pass come through this loss only
    (the failed pass's accumulation is undone by restoring what .grad held when backward began).  A forward pass whose backward
    never runs is checked at the next forward call.  False (or PLMC_LATE_CHECK=0): the check sits at the end of the forward
    pass, as in round 3.  Measured, metric shape: one latent per rank 5.32 -> 4.81 ms per step (the host used to sit out the
    3.5 ms sweep before queueing ~110 launches of loss terms and backward pass); q = 8: 17.85 -> 17.81."""
    _stack = [True]

    @classmethod
    def on(cls):
        return cls._stack[-1]


class prediction_cache(_Value):
    """prediction_cache("lazy" | "eager" | "off"): when an eval-mode model builds the factorisation it keeps between prediction
    calls (_engine.PosteriorCache; gpytorch's prediction strategy keeps its own from the first call).  "lazy" (default): on the
    second call with unchanged parameters and data -- a single prediction costs and holds no more than it did without a cache;
    "eager": on the first call; "off": never."""
    _stack = ["lazy"]

    def __init__(self, value):
        if value not in ("lazy", "eager", "off"):
            raise ValueError("prediction_cache: 'lazy', 'eager' or 'off'")
        super().__init__(value)

    @classmethod
    def value(cls):
        return cls._stack[-1]
