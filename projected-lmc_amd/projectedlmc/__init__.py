"""projectedlmc -- MI355X-native drop-in for the reference package of the same name.

Same public names as /root/reference/projectedlmc/__init__.py (`from .projected_lmc import *`),
with the gpytorch-dependent hot path replaced by hand-written HIP kernels (libplmc_hip.so,
C ABI in include/plmc.h).  gpytorch is not a dependency: the few of its types the reference's
users touch (kernels, means, likelihoods, distributions, mlls, settings, constraints) are
provided here as thin modules with the same names and parameter layout.
"""
from . import settings, constraints, kernels, means, likelihoods, distributions, mlls, parallel, priors  # noqa: F401
from .priors import NormalPrior, MultivariateNormalPrior  # noqa: F401
from .kernels import RBFKernel, MaternKernel, SplineKernel, ScaleKernel, MultitaskKernel, LCMKernel, IndexKernel  # noqa: F401
from .means import ZeroMean, ConstantMean, MultitaskMean, LinearMean, PolynomialMean  # noqa: F401
from .likelihoods import GaussianLikelihood, MultitaskGaussianLikelihood  # noqa: F401
from .distributions import MultivariateNormal, MultitaskMultivariateNormal  # noqa: F401
from .mlls import ExactMarginalLogLikelihood  # noqa: F401
from .models import (ExactGPModel, handle_covar_, init_lmc_coefficients, ScalarParam,  # noqa: F401
                     PositiveDiagonalParam, UpperTriangularParam, LowerTriangularParam)
from .projected import LMCMixingMatrix, ProjectedGPModel, ProjectedLMCmll  # noqa: F401
from .multitask import MultitaskGPModel  # noqa: F401
from .sgpr import InducingPointKernel  # noqa: F401
from . import variational  # noqa: F401
from .variational import (VariationalMultitaskGPModel, CustomLMCVariationalStrategy, VariationalELBO,  # noqa: F401
                          CholeskyVariationalDistribution, VariationalStrategy, LMCVariationalStrategy)
