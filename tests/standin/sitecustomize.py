"""TEST INFRASTRUCTURE ONLY (tests/test_bench_launch.py puts this directory on PYTHONPATH of the processes it starts).

With PLMC_TEST_STANDIN=1 the n x n log-likelihood call of the product -- which needs the GPU -- is replaced by the
oracle's torch implementation, exactly as tests/test_sharding_gloo.py does inside its workers, so that the LAUNCH path of
bench.py (self-launch, rendezvous, latent sharding, fused all-reduce, JSON line) can be rehearsed on a box without a GPU.
Nothing in the product or in a measured run ever imports this file."""
import os
import sys

if os.environ.get("PLMC_TEST_STANDIN") == "1":
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for p in (ROOT, os.path.join(ROOT, "projected-lmc_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        import projectedlmc  # noqa: F401
        from projectedlmc import _engine, distributions
        from oracle import gp_math as gm

        _KINDS = {"rbf": ("rbf", 2.5), "matern12": ("matern", 0.5), "matern32": ("matern", 1.5), "matern52": ("matern", 2.5)}

        def _standin_log_prob(kind, X, ell, oscale, noise, y, hyper=None):
            k, nu = _KINDS[kind]
            return gm.exact_latent_log_prob(k, X, ell, noise, y, oscale, nu)

        _engine.exact_latent_log_prob = _standin_log_prob
        distributions.MultivariateNormal.log_prob = lambda self, value: _standin_log_prob(
            self._covar.kind, self._covar.x1, self._covar.ell, self._covar.oscale, self._covar.noise.reshape(-1),
            (value - self.loc).reshape(self._covar.ell.shape[0], -1))
    except Exception as exc:                       # the launcher process itself may not need (or have) the package
        sys.stderr.write("[standin] not installed: %r\n" % (exc,))
