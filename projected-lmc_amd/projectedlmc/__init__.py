"""projectedlmc -- MI355X-native drop-in for the reference package of the same name.

Same public names as /root/reference/projectedlmc/__init__.py (`from .projected_lmc import *`),
with the gpytorch-dependent hot path replaced by hand-written HIP kernels (libplmc_hip.so)."""
from . import settings  # noqa: F401
