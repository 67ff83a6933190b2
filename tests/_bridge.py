"""Test helper: the product-model -> oracle-dict bridge lives in oracle/bridge.py (bench.py's cpu_baseline leg uses it too)."""
from oracle.bridge import KIND, oracle_params, param_map, perturb_   # noqa: F401
