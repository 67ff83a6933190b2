"""Generates tests/golden/third_party_v1.json -- run from the repo root:  python tests/golden/make_third_party.py

INDEPENDENT cross-check of the restated arithmetic.  The reference's own third-party arithmetic
(gpytorch 1.11) cannot be installed offline and the reference ships no fixtures, so parity stays "unpinned" by rule;
what this closes is the RESTATEMENT risk of projected_lmc.py:151-157 (ARD RBF / Matern-5/2 kernels) and :1200-1201
(K + sigma^2 I, dense Gaussian log-density and its gradient): the expected numbers below come from two other
implementations of the same mathematics that ARE in this container and share no code with oracle/ or the HIP path:

  * scikit-learn  GaussianProcessRegressor.log_marginal_likelihood(theta, eval_gradient=True) with
    ConstantKernel * RBF(length_scale=[...]) + WhiteKernel  and  ConstantKernel * Matern(nu=2.5, ...) + WhiteKernel
    (value + gradient w.r.t. outputscale, every lengthscale, noise);
  * scipy.stats.multivariate_normal.logpdf on a covariance assembled with numpy.kron from scikit-learn kernel
    matrices: the dense LMC density  log N(vec Y; 0, sum_i K_i (x) B_i + I (x) Sigma)  (data-major interleaving).

Only numbers are stored (inputs, natural parameters, expected outputs); nothing of either library travels."""
import json
import os

import numpy as np
import scipy
import sklearn
from scipy.stats import multivariate_normal
from sklearn.gaussian_process import GaussianProcessRegressor
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "third_party_v1.json")


def gpr_case(name, kind, n, d, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(-1, 1, (n, d))
    y = np.sin(X @ rng.normal(size=d)) + 0.3 * rng.normal(size=n)
    ell = rng.uniform(0.4, 1.3, d)
    os_, noise = float(rng.uniform(0.6, 1.8)), float(rng.uniform(0.05, 0.4))
    base = RBF(length_scale=ell) if kind == "rbf" else Matern(length_scale=ell, nu=2.5)
    kernel = ConstantKernel(os_) * base + WhiteKernel(noise)
    gpr = GaussianProcessRegressor(kernel=kernel, optimizer=None, alpha=0.0).fit(X, y)
    theta = gpr.kernel_.theta                                   # log(os), log(ell_1..d), log(noise)
    lml, glog = gpr.log_marginal_likelihood(theta, eval_gradient=True)
    nat = np.exp(theta)
    g = glog / nat                                              # d/d theta = (d/d log theta) / theta
    assert np.allclose(nat, np.concatenate([[os_], ell, [noise]]))
    return {"kind": "gpr", "name": name, "kernel": kind, "X": X.tolist(), "y": y.tolist(), "ell": ell.tolist(),
            "outputscale": os_, "noise": noise, "log_marginal_likelihood": float(lml),
            "grad_outputscale": float(g[0]), "grad_ell": g[1:1 + d].tolist(), "grad_noise": float(g[-1])}


def lmc_case(name, kind, n, d, p, q, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(-1, 1, (n, d))
    Y = rng.normal(size=(n, p))
    ell = rng.uniform(0.4, 1.2, (q, d))
    F = rng.normal(size=(q, p, 1))
    v = rng.uniform(0.05, 0.3, (q, p))
    B = np.stack([F[i] @ F[i].T + np.diag(v[i]) for i in range(q)])
    Sigma = np.diag(rng.uniform(0.05, 0.3, p))
    C = np.kron(np.eye(n), Sigma)
    for i in range(q):
        k = RBF(length_scale=ell[i]) if kind == "rbf" else Matern(length_scale=ell[i], nu=2.5)
        C = C + np.kron(k(X), B[i])                             # data-major: index = i_point * p + i_task
    lp = multivariate_normal(mean=np.zeros(n * p), cov=C).logpdf(Y.reshape(-1))
    return {"kind": "lmc", "name": name, "kernel": kind, "X": X.tolist(), "Y": Y.tolist(), "ell": ell.tolist(),
            "covar_factor": F.tolist(), "var": v.tolist(), "B": B.tolist(), "Sigma": Sigma.tolist(), "log_density": float(lp)}


def main():
    cases = [gpr_case("sklearn_rbf_ard_n60_d3", "rbf", 60, 3, 0), gpr_case("sklearn_matern52_ard_n80_d4", "matern", 80, 4, 1),
             gpr_case("sklearn_matern52_ard_n300_d8", "matern", 300, 8, 2), gpr_case("sklearn_rbf_ard_n257_d2", "rbf", 257, 2, 3),
             lmc_case("scipy_lmc_rbf_n40_p3_q2", "rbf", 40, 2, 3, 2, 4), lmc_case("scipy_lmc_matern52_n50_p4_q3", "matern", 50, 3, 4, 3, 5)]
    json.dump({"generator": "tests/golden/make_third_party.py", "scikit_learn": sklearn.__version__, "scipy": scipy.__version__,
               "numpy": np.__version__, "cases": cases}, open(OUT, "w"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
