"""Oracle: read a product ProjectedGPModel's raw parameters (state-dict names of projected_lmc.py:897-993) into the oracle's dict
form (oracle/projected.py).  TEST INFRASTRUCTURE ONLY: used by tests/ and by bench.py's cpu_baseline leg; it touches the product
model through its public state dict only and is never imported by the product."""
import math

import torch

KIND = {"rbf": ("rbf", 2.5), "matern12": ("matern", 0.5), "matern32": ("matern", 1.5), "matern52": ("matern", 2.5)}


def oracle_params(model, dtype=torch.float64):
    cm = model.covar_module
    base = cm.base_kernel if hasattr(cm, "base_kernel") else cm
    kind, nu = KIND[base.kind]
    lb = model.likelihood.noise_covar.raw_noise_constraint.lower_bound
    sd = {k: v.detach().cpu().to(dtype) for k, v in model.state_dict().items()}
    P = dict(kind=kind, nu=nu, n_tasks=model.n_tasks, n_latents=model.n_latents,
             mode=model.lmc_coefficients.mode, BDN=not hasattr(model, "M"), eps=model.eps,
             scalar_B=model.scalar_B, diagonal_B=model.diagonal_B, noise_lb=lb, noise_thresh=math.log(lb),
             bulk=model.lmc_coefficients.bulk,
             raw_noise=sd["likelihood.noise_covar.raw_noise"],
             raw_lengthscale=sd["covar_module.base_kernel.raw_lengthscale" if hasattr(cm, "base_kernel")
                                else "covar_module.raw_lengthscale"],
             raw_outputscale=sd.get("covar_module.raw_outputscale"))
    if model.lmc_coefficients.bulk:
        P["H"] = sd["lmc_coefficients.H"]
    else:
        # bulk=False (projected_lmc.py:963-970): torch's orthogonal parametrisation of Q_plus (original + the fixed base of its
        # trivialisation) and the parametrised R
        from torch.nn.utils import parametrize
        lmc = model.lmc_coefficients
        orth = lmc.parametrizations.Q_plus[0]
        P["Q_plus_original"] = sd["lmc_coefficients.parametrizations.Q_plus.original"]
        P["Q_plus_base"] = sd.get("lmc_coefficients.parametrizations.Q_plus.0.base")
        P["ortho_param"] = orth.orthogonal_map.name
        P["R_original"] = sd["lmc_coefficients.parametrizations.R.original"]
        P["diagonal_R"] = type(lmc.parametrizations.R[0]).__name__ == "PositiveDiagonalParam"
    if "parametrizations.log_B_tilde.original" in sd:
        P["log_B_tilde"] = sd["parametrizations.log_B_tilde.original"]
    elif "log_B_tilde" in sd:
        P["log_B_tilde"] = sd["log_B_tilde"]
    if "parametrizations.B_tilde_inv_chol.original" in sd:
        P["B_tilde_inv_chol_raw"] = sd["parametrizations.B_tilde_inv_chol.original"]
    if "M" in sd:
        P["M"] = sd["M"]
    return P


# product parameter name -> oracle dict key
def param_map(model):
    cm = model.covar_module
    m = {"lmc_coefficients.H": "H", "lmc_coefficients.parametrizations.Q_plus.original": "Q_plus_original",
         "lmc_coefficients.parametrizations.R.original": "R_original", "likelihood.noise_covar.raw_noise": "raw_noise", "M": "M",
         "parametrizations.log_B_tilde.original": "log_B_tilde", "log_B_tilde": "log_B_tilde",
         "parametrizations.B_tilde_inv_chol.original": "B_tilde_inv_chol_raw"}
    if hasattr(cm, "base_kernel"):
        m["covar_module.base_kernel.raw_lengthscale"] = "raw_lengthscale"
        m["covar_module.raw_outputscale"] = "raw_outputscale"
    else:
        m["covar_module.raw_lengthscale"] = "raw_lengthscale"
    return m


def perturb_(model, seed=1, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.add_(scale * torch.randn(p.shape, generator=g, dtype=torch.float64).to(p.dtype).to(p.device))
            if name.endswith("B_tilde_inv_chol.original"):
                p.copy_(p.tril())
    return model
