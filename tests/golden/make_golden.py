"""Generates tests/golden/projected_v1.json -- run from the repo root:  python tests/golden/make_golden.py

Golden vectors for the hot path.  The reference cannot be imported here (its third-party arithmetic,
gpytorch 1.11, is not installable offline: SURVEY.md 8c) and ships no fixtures of its own, so these are
outputs of the fp64 CPU restatement in oracle/ on small explicit inputs -- PARITY UNPINNED, as the oracle
header says -- cross-validated by the reference-independent checks of tests/test_oracle_selfcheck.py.
They pin (a) the oracle against drift (CPU test) and (b) the HIP path against fixed numbers (GPU test).
Everything a case needs is stored explicitly (inputs, raw parameters by state-dict name, expected
outputs); no RNG is involved in reading them back."""
import json
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd"), os.path.join(ROOT, "tests")]
import projectedlmc as plmc                      # noqa: E402  (host-side model construction only: no device code runs)
from oracle import projected as pj               # noqa: E402
from oracle import gp_math as gm                 # noqa: E402
from _bridge import oracle_params, param_map, perturb_   # noqa: E402

VARIANTS = {
    "PLMC": dict(BDN=False, diagonal_B=False, scalar_B=False),
    "PLMC_fast": dict(BDN=True, diagonal_B=True, scalar_B=True),
    "PLMC_diagB": dict(BDN=False, diagonal_B=True, scalar_B=False),
    # bulk=False (projected_lmc.py:963-970): Q_plus orthogonally parametrised, R parametrised; "oilmm" = realdata_experiments.py:107-111
    "oilmm_nonbulk": dict(BDN=True, diagonal_B=True, scalar_B=True, diagonal_R=True, bulk=False),
    "PLMC_nonbulk": dict(BDN=False, diagonal_B=False, scalar_B=False, bulk=False),
}


def tolist(t):
    return t.detach().to(torch.float64).cpu().tolist()


def projected_case(name, variant, kernel, oscale, n, d, p, q, ns, seed):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(n, p, generator=g, dtype=torch.float64)
    Xs = 2 * torch.rand(ns, d, generator=g, dtype=torch.float64) - 1
    torch.manual_seed(seed + 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = plmc.ProjectedGPModel(X, Y, p, q, mean_type=plmc.ZeroMean, kernel_type=getattr(plmc, kernel),
                                  init_lmc_coeffs=True, outputscales=oscale, **VARIANTS[variant])
    m = perturb_(m.double(), seed=seed + 2)
    P = oracle_params(m)
    for k in pj.tensor_keys(P):
        P[k].requires_grad_(True)
    loss = -pj.projected_mll(P, X, Y)
    loss.backward()
    pm = param_map(m)
    grads = {pname: tolist(P[pm[pname]].grad) for pname, _ in m.named_parameters() if P[pm[pname]].grad is not None}
    with torch.no_grad():
        mean, cov = pj.task_posterior(P, X, Y, Xs)
        _, var_obs = pj.observed_posterior(P, X, Y, Xs)
    return dict(name=name, kind="projected", variant=variant, ctor=VARIANTS[variant], kernel=kernel, outputscales=oscale,
                n_tasks=p, n_latents=q, X=tolist(X), Y=tolist(Y), Xs=tolist(Xs),
                params={k: tolist(v) for k, v in m.named_parameters()},
                # buffers that are not a function of the constructor arguments: the base of the orthogonal trivialisation
                # (torch completes a rectangular Q_plus with a random block)
                buffers={k: tolist(v) for k, v in m.named_buffers() if k.endswith(".base")},
                loss=float(loss.detach()), grads=grads, pred_mean=tolist(mean),
                pred_var=tolist(torch.diagonal(cov).reshape(ns, p)), pred_var_observed=tolist(var_obs))


def exact_case(name, kernel, kind, n, d, ns, seed):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    y = torch.sin(3 * X[:, 0]) + 0.2 * torch.randn(n, generator=g, dtype=torch.float64)
    Xs = 2 * torch.rand(ns, d, generator=g, dtype=torch.float64) - 1
    raw_ls = (0.4 * torch.randn(1, d, generator=g, dtype=torch.float64)).requires_grad_(True)
    raw_noise = torch.tensor([-1.5], dtype=torch.float64, requires_grad=True)
    lb = 1e-4                                                   # GaussianLikelihood default lower bound [gpytorch-knowledge]
    ell, noise = gm.softplus(raw_ls), gm.softplus(raw_noise) + lb
    mll = gm.exact_latent_log_prob(kind, X, ell, noise, y[None], None, 2.5)[0] / n
    mll.backward()
    with torch.no_grad():
        mu, cov = gm.exact_gp_posterior(kind, X, ell, noise, y[None], Xs, None, 2.5)
    return dict(name=name, kind="exact", kernel=kernel, X=tolist(X), y=tolist(y), Xs=tolist(Xs), noise_lower_bound=lb,
                raw_lengthscale=tolist(raw_ls), raw_noise=tolist(raw_noise), mll=float(mll.detach()),
                grad_raw_lengthscale=tolist(raw_ls.grad), grad_raw_noise=tolist(raw_noise.grad),
                pred_mean=tolist(mu[0]), pred_var=tolist(torch.diagonal(cov[0])))


def svd_case(seed):
    g = torch.Generator().manual_seed(seed)
    Y = torch.randn(40, 6, generator=g, dtype=torch.float64) @ torch.randn(6, 6, generator=g, dtype=torch.float64)
    return dict(name="svd_init", kind="svd", Y=tolist(Y), n_latents=3, coeffs=tolist(pj.svd_init(Y, 3)))


if __name__ == "__main__":
    cases = [
        projected_case("plmc_fast_matern", "PLMC_fast", "MaternKernel", False, 96, 3, 5, 2, 7, 100),
        projected_case("plmc_full_rbf_outputscale", "PLMC", "RBFKernel", True, 90, 2, 4, 3, 6, 200),
        projected_case("plmc_diagB_matern", "PLMC_diagB", "MaternKernel", False, 70, 2, 4, 2, 5, 300),
        exact_case("exact_rbf", "RBFKernel", "rbf", 80, 2, 6, 400),
        exact_case("exact_matern", "MaternKernel", "matern", 72, 3, 5, 500),
        svd_case(600),
    ]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "projected_v1.json")
    if "--nonbulk-only" not in sys.argv:
        json.dump(dict(version=1, generator="tests/golden/make_golden.py", cases=cases), open(out, "w"))
        print("wrote", out, os.path.getsize(out), "bytes")
    # round 4: the separately parametrised mixing matrix (bulk=False), in a file of its own so that v1 stays byte-identical
    cases2 = [
        projected_case("oilmm_nonbulk_matern", "oilmm_nonbulk", "MaternKernel", False, 88, 3, 5, 2, 6, 700),
        projected_case("plmc_nonbulk_rbf_outputscale", "PLMC_nonbulk", "RBFKernel", True, 80, 2, 4, 2, 5, 800),
    ]
    out2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "projected_nonbulk_v1.json")
    json.dump(dict(version=1, generator="tests/golden/make_golden.py", cases=cases2), open(out2, "w"))
    print("wrote", out2, os.path.getsize(out2), "bytes")
