"""CPU oracle for the projected-LMC exact-GP hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain torch-CPU (fp64 by default) restatement of the arithmetic
the reference obtains from gpytorch / linear_operator for the path named in
BASELINE.json (covariance assembly -> dense Cholesky MLL + gradient -> LMC mixing /
projection).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / reported baseline.  The product
package (`projected-lmc_amd/projectedlmc`) never imports it and has no CPU fallback.

PARITY UNPINNED.  The reference (`/root/reference/projectedlmc/projected_lmc.py`)
imports gpytorch==1.11 / linear_operator==0.5.0 at module top (lines 4, 13-16);
neither is installed in this container nor fetchable offline, and
the reference ships no tests, golden vectors or fixtures.  The oracle
is therefore pinned only by reference-independent checks (tests/test_oracle_*.py):
finite-difference gradients, closed-form kernel values, the identity
n * ProjectedLMCmll == dense log N(vec Y; 0, sum_i K_i (x) h_i h_i^T + I (x) Sigma),
batch == loop equivalence, and the p == q edge case.  gpytorch semantics restated
from knowledge of the pinned versions are marked "[gpytorch-knowledge]".
"""
from . import gp_math, projected, lmc_dense  # noqa: F401
