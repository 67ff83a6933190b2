"""Oracle: the Projected-LMC model algebra and its loss.  TEST INFRASTRUCTURE ONLY.

Functional restatement (explicit tensors in, tensors out; torch-CPU, autograd-able) of

* `init_lmc_coefficients`                           projected_lmc.py:183-201
* parametrisations ScalarParam / UpperTriangularParam / LowerTriangularParam   :207-258
* `LMCMixingMatrix.QR` / `.forward`, bulk (:864-872, :877-882) and separately parametrised
  `Q_plus` . `R` (`bulk=False`: :851-853, :873, :884, :963-970; loss branch :1237)
* `ProjectedGPModel.__init__` initial values                                  :916-993
* `.projection_matrix` :1003-1012, `.project_data` :1014-1021,
  `.full_likelihood` (task-noise Sigma) :1023-1060, `.B_tilde` :1076-1086
* `ProjectedGPModel.__call__` eval branch (task posterior)                    :1133-1155
* `ProjectedLMCmll.forward`                                                   :1178-1241

A model is a plain dict `P` of tensors/flags (see `init_params`).  Nothing here is
used by the product path.
"""
import math
import numpy as np
import torch

from . import gp_math as gm


# ----------------------------------------------------------------------------- init
def svd_init(Y, n_latents, QR_form=False):
    """projected_lmc.py:183-201.  Uses sklearn's randomized_svd exactly as the reference
    does (third-party dependency of the reference, present in this container)."""
    from sklearn.utils.extmath import randomized_svd
    n_data, n_tasks = Y.shape
    if n_data >= n_latents:
        U, S, _ = randomized_svd(Y.cpu().numpy().T, n_components=n_latents, random_state=0)
        U, S = torch.as_tensor(U, dtype=Y.dtype), torch.as_tensor(S, dtype=Y.dtype)
    else:
        Q, R = np.linalg.qr(Y.cpu().numpy().T, mode="complete")
        S = 1e-3 * torch.ones(n_latents, dtype=Y.dtype)
        S[:n_data] = torch.as_tensor(np.diag(R).copy(), dtype=Y.dtype)
        U = torch.as_tensor(Q[:, :n_latents], dtype=Y.dtype)
    if QR_form:
        return U, S
    return (U * S / np.sqrt(n_data - 1)).T


def init_params(X, Y, n_latents, *, kind="rbf", nu=2.5, init_lmc_coeffs=False, BDN=True,
                diagonal_B=False, scalar_B=False, noise_thresh=-9.0, noise_init=1e-2,
                outputscales=False, fake_coeffs=None, eps=1e-3, bulk=True, diagonal_R=False,
                ortho_param="matrix_exp", completion=None):
    """Initial parameter dict of a ProjectedGPModel (projected_lmc.py:916-993).
    `fake_coeffs` (p x q) replaces the reference's torch.randn draw (:955) so that
    product and oracle can be seeded identically.  `bulk=False` (:963-970): the values torch's
    `register_parametrization` leaves in `original` -- -Id for the trivialised orthogonal map with `base` = Q_plus
    completed to p x p (`completion`: the p x (p-c) block torch draws with randn for a rectangular Q_plus; any
    completion gives the same Q_plus at initialisation), the log-diagonal right inverses for R (:224-227, :236-240)."""
    n, p = Y.shape
    q = n_latents
    d = X.shape[1]
    dt = Y.dtype
    if init_lmc_coeffs:
        if scalar_B and BDN:
            Q_plus, R = svd_init(Y, q, QR_form=True)                       # :933
        else:
            Q_plus, R_padded = svd_init(Y, p, QR_form=True)                # :935
            R = R_padded[:q]
    else:
        if fake_coeffs is None:
            fake_coeffs = torch.randn(p, q, dtype=dt)
        Q_plus, R_padded, _ = torch.linalg.svd(fake_coeffs)                # :956
        R = R_padded[:q]
        if scalar_B and BDN:
            Q_plus = Q_plus[:, :q]                                         # :960
    R = torch.diag_embed(R) / np.sqrt(n - 1)                               # :962
    if Q_plus.shape[1] == Q_plus.shape[0]:                                 # :832-849
        mode = "Q_plus"
        R_padded = torch.eye(p, dtype=dt)
        R_padded[:q, :q] = R
        H = Q_plus @ R_padded
    else:
        mode = "Q"
        H = Q_plus @ R
    P = dict(kind=kind, nu=nu, n_tasks=p, n_latents=q, mode=mode, BDN=BDN, eps=eps,
             scalar_B=scalar_B, diagonal_B=(diagonal_B or scalar_B), noise_lb=math.exp(noise_thresh),
             noise_thresh=noise_thresh, bulk=bulk,
             raw_lengthscale=torch.zeros(q, 1, d, dtype=dt),
             raw_outputscale=(torch.zeros(q, dtype=dt) if outputscales else None),
             raw_noise=torch.zeros(q, 1, dtype=dt))
    if bulk:
        P["H"] = H.clone()
    else:
        c = Q_plus.shape[1]
        P.update(diagonal_R=diagonal_R, ortho_param=ortho_param)
        if ortho_param == "householder":                                   # no trivialisation: geqrf reflectors, signed diagonal
            A, tau = torch.geqrf(Q_plus)
            A.diagonal().sign_()
            A.diagonal()[tau == 0.0] *= -1
            P["Q_plus_original"], P["Q_plus_base"] = A, None
        else:
            Qc = Q_plus.clone()
            if c != p:
                N = completion if completion is not None else torch.randn(p, p - c, dtype=dt)
                Qc = torch.cat([Qc, N], dim=-1)
            Qf, Rf = torch.linalg.qr(Qc)                                   # torch's _make_orthogonal
            P["Q_plus_base"] = Qf * torch.diagonal(Rf).sgn().unsqueeze(-2)
            negId = torch.zeros(p, c, dtype=dt)
            negId.diagonal().fill_(-1.0)
            P["Q_plus_original"] = negId
        Rl = R.clone()
        Rl.diagonal().copy_(torch.log(torch.diagonal(R)))
        P["R_original"] = torch.diag_embed(torch.diagonal(Rl)) if diagonal_R else Rl
    if scalar_B or diagonal_B:
        P["log_B_tilde"] = math.log(noise_init) * torch.ones(p - q, dtype=dt)      # :975/:980
    else:
        # :983-984 -- the parameter is created as diag(log(1/noise_init)) and THEN parametrised;
        # torch stores original = right_inverse(value) (:255-258: log of the diagonal), so the
        # effective initial B_tilde_inv_chol is diag(log(1/noise_init)), not its exponential.
        P["B_tilde_inv_chol_raw"] = torch.diag_embed(math.log(math.log(1.0 / noise_init)) * torch.ones(p - q, dtype=dt))
    if not BDN:
        P["M"] = torch.zeros(q, p - q, dtype=dt)                                    # :988
    return P


def tensor_keys(P):
    """The parameters (what an optimiser steps): every tensor but the fixed base of the orthogonal trivialisation (a buffer)."""
    return [k for k, v in P.items() if torch.is_tensor(v) and k != "Q_plus_base"]


# ----------------------------------------------------------------- constrained views
def lengthscale(P):
    return gm.softplus(P["raw_lengthscale"]).reshape(P["n_latents"], -1)          # (q,d)


def outputscale(P):
    return None if P.get("raw_outputscale") is None else gm.softplus(P["raw_outputscale"])


def projected_noise(P):
    """:996-1000 ; GaussianLikelihood noise = softplus(raw)+exp(noise_thresh) (:920-921)."""
    return gm.softplus(P["raw_noise"]).reshape(-1) + P["noise_lb"]


def log_B_tilde(P):
    """ScalarParam (:207-218) with bounds (noise_thresh,-noise_thresh) when scalar_B (:976);
    GreaterThan(noise_thresh) constraint registered but *not applied* by attribute access
    when only diagonal_B (:980-981) [gpytorch-knowledge: register_constraint on a parameter
    named without the raw_ prefix leaves plain attribute reads untransformed]."""
    lb = P["log_B_tilde"]
    if P["scalar_B"]:
        if lb.numel() == 0:
            return lb
        return torch.ones_like(lb) * torch.clamp(lb.mean(), P["noise_thresh"], -P["noise_thresh"])
    return lb


def B_tilde_inv_chol(P):
    """LowerTriangularParam.forward (:250-254): tril with exp(clamp(diag))."""
    X = P["B_tilde_inv_chol_raw"]
    lower = X.tril()
    dg = torch.exp(torch.clamp(torch.diagonal(lower), P["noise_thresh"], -P["noise_thresh"]))
    return lower - torch.diag_embed(torch.diagonal(lower)) + torch.diag_embed(dg)


def orthogonal_map(X, base, ortho_param="matrix_exp"):
    """What `torch.nn.utils.parametrizations.orthogonal(lmc, name="Q_plus", orthogonal_map=ortho_param,
    use_trivialization=(ortho_param != 'householder'))` (:963-965) evaluates for a tall or square p x c matrix --
    third-party (torch) arithmetic the reference calls, restated from its published definition
    (torch/nn/utils/parametrizations.py, `_Orthogonal.forward`):
      matrix_exp / cayley: A = L - L^T with L = tril(X) padded to p x p; Q = expm(A) or (I - A/2)^-1 (I + A/2), first c columns;
      householder: product of the reflectors in tril(X, -1), columns signed by diag(X);
    with the trivialisation (`base` p x p, fixed at registration; `original` starts at -Id) Q <- base @ Q."""
    n, k = X.shape
    assert n >= k
    if ortho_param in ("matrix_exp", "cayley"):
        L = X.tril()
        if n != k:
            L = torch.cat([L, L.new_zeros(n, n - k)], dim=-1)
        A = L - L.T
        if ortho_param == "matrix_exp":
            Q = torch.matrix_exp(A)
        else:
            Id = torch.eye(n, dtype=A.dtype)
            Q = torch.linalg.solve(Id - 0.5 * A, Id + 0.5 * A)
        Q = Q[:, :k]
    else:
        A = X.tril(diagonal=-1)
        tau = 2.0 / (1.0 + (A * A).sum(dim=-2))
        Q = torch.linalg.householder_product(A, tau)
        Q = Q * torch.diagonal(X).detach().int().unsqueeze(-2)
    if base is not None:
        Q = base @ Q
    return Q


def R_param(P):
    """The parametrised R of `bulk=False` (:966-970): PositiveDiagonalParam (:220-227, `diagonal_R`) keeps exp of the
    diagonal of `original` and nothing else; UpperTriangularParam (:229-240) its upper triangle with exp on the diagonal."""
    X = P["R_original"]
    if P["diagonal_R"]:
        return torch.diag_embed(torch.exp(torch.diagonal(X)))
    upper = X.triu()
    return upper - torch.diag_embed(torch.diagonal(upper)) + torch.diag_embed(torch.exp(torch.diagonal(upper)))


def Q_plus(P):
    """`lmc_coefficients.Q_plus` of `bulk=False`: the orthogonal parametrisation of `parametrizations.Q_plus.original`."""
    return orthogonal_map(P["Q_plus_original"], P.get("Q_plus_base"), P.get("ortho_param", "matrix_exp"))


def QR(P):
    """LMCMixingMatrix.QR (:864-875): bulk mode re-factors H (:865-872); otherwise (Q, R, Q_orth) are read off the
    parametrised `Q_plus` and `R` (:873, :855-862)."""
    q = P["n_latents"]
    if not P.get("bulk", True):
        Qp = Q_plus(P)
        if P["mode"] == "Q_plus":
            return Qp[:, :q], R_param(P), Qp[:, q:]
        return Qp, R_param(P), Qp[:, q:]                                   # :861-862: p x 0 in mode 'Q'
    Q_plus_, R_padded = torch.linalg.qr(P["H"])
    if P["mode"] == "Q_plus":
        return Q_plus_[:, :q], R_padded[:q, :q], Q_plus_[:, q:]
    return Q_plus_, R_padded, None


def lmc_coefficients(P):
    """LMCMixingMatrix.forward (:877-884): q x p."""
    if not P.get("bulk", True):
        Q, R, _ = QR(P)
        return (Q @ R).T                                                   # :884
    if P["mode"] == "Q":
        return P["H"].T
    return P["H"][:, :P["n_latents"]].T


def projection_matrix(P):
    """:1003-1012 -> T (p x q)."""
    Q, R, Q_orth = QR(P)
    H_pinv = torch.linalg.solve_triangular(R.T, Q, upper=False, left=False)
    if "M" in P:
        return H_pinv + Q_orth @ P["M"].T * projected_noise(P)[None, :]
    return H_pinv


def project_data(P, Y):
    """:1014-1021 -> (q x n)."""
    Q, R, Q_orth = QR(P)
    out = torch.linalg.solve_triangular(R, Q.T @ Y.T, upper=True)
    if "M" in P:
        out = out + projected_noise(P)[:, None] * P["M"] @ Q_orth.T @ Y.T
    return out


def B_tilde(P):
    """:1076-1086."""
    if P["diagonal_B"]:
        return torch.diag_embed(torch.exp(log_B_tilde(P)))
    pq = P["n_tasks"] - P["n_latents"]
    L_inv = torch.linalg.solve_triangular(B_tilde_inv_chol(P), torch.eye(pq, dtype=P["raw_noise"].dtype), upper=False)
    return L_inv.T @ L_inv


def full_noise_covariance(P):
    """Sigma (p x p) of `full_likelihood` before jitter (:1023-1060)."""
    p, q = P["n_tasks"], P["n_latents"]
    dt = P["raw_noise"].dtype
    Q, R, Q_orth = QR(P)
    QRm = Q @ R
    sp = projected_noise(P)
    if "M" in P:
        if P["diagonal_B"]:
            Broot = torch.diag_embed(torch.exp(log_B_tilde(P) / 2))
        else:
            Broot = torch.linalg.solve_triangular(B_tilde_inv_chol(P), torch.eye(p - q, dtype=dt), upper=False).T
        Bt = Broot @ Broot.T
        B_term = Q_orth @ Bt @ Q_orth.T
        M_term = -QRm @ (sp[:, None] * P["M"]) @ Bt @ Q_orth.T
        D_rot = torch.diag_embed(sp) + sp[:, None] * P["M"] @ Bt @ P["M"].T * sp[None, :]
        return QRm @ D_rot @ QRm.T + M_term + M_term.T + B_term
    if P["scalar_B"]:
        lb = log_B_tilde(P)
        B_term = torch.exp(lb[0]) * (torch.eye(p, dtype=dt) - Q @ Q.T) if lb.numel() > 0 else 0.0
    else:
        if P["diagonal_B"]:
            Broot = torch.diag_embed(torch.exp(log_B_tilde(P) / 2))
        else:
            Broot = torch.linalg.solve_triangular(B_tilde_inv_chol(P), torch.eye(p - q, dtype=dt), upper=False).T
        Br = Q_orth @ Broot
        B_term = Br @ Br.T
    Droot = QRm * torch.sqrt(sp)[None, :]
    return Droot @ Droot.T + B_term


def full_noise_factor(P):
    """Jittered Cholesky written into task_noise_covar_factor (:1063-1072)."""
    Sigma = full_noise_covariance(P).detach()
    eps = 1e-6
    eye = torch.eye(P["n_tasks"], dtype=Sigma.dtype)
    while eps < P["eps"]:
        try:
            return torch.linalg.cholesky(Sigma + eps * eye)
        except Exception:
            eps *= 10
    raise RuntimeError("full noise covariance not PD")


# ------------------------------------------------------------------------- the loss
def projection_terms(P, Y):
    """The three `proj_term_list` entries + constant of ProjectedLMCmll.forward
    (:1205-1238).  Returns (list_of_3, constant)."""
    p, q = P["n_tasks"], P["n_latents"]
    n = Y.shape[0]
    Q, R, Q_orth = QR(P)
    terms = [0, 0, 0]
    if "M" not in P and P["scalar_B"]:
        lb = log_B_tilde(P)
        if lb.numel() > 0:
            Y2 = (Y ** 2).sum()                                           # buffer Y_squared_norm :978
            terms[1] = -0.5 * torch.exp(-lb[0]) * (Y2 - (Y @ Q).pow(2).sum()) / n      # :1215
            root_diag = lb / 2
        else:
            terms[1] = 0.0
            root_diag = torch.zeros(1, dtype=Y.dtype)
    else:
        if P["diagonal_B"]:
            lb = log_B_tilde(P)
            root_diag = lb / 2
            rot = Y @ Q_orth
            # reference forms the n x n matrix and takes its trace (:1224,:1230)
            terms[1] = -0.5 * ((rot * torch.exp(-lb)[None, :]) * rot).sum() / n
        else:
            Bc = B_tilde_inv_chol(P)
            root_diag = -torch.log(torch.diagonal(Bc))
            root = Y @ Q_orth @ Bc
            terms[1] = -0.5 * (root * root).sum() / n
    terms[0] = -0.5 * 2 * torch.sum(root_diag)                           # :1233
    if P.get("bulk", True):
        terms[2] = -0.5 * torch.log(torch.diagonal(R) ** 2).sum()        # :1235 (bulk)
    else:
        terms[2] = -0.5 * 2 * torch.diagonal(P["R_original"]).sum()      # :1237 (log of the diagonal IS the raw diagonal)
    const = -0.5 * (p - q) * math.log(2 * math.pi)                       # :1238
    return terms, const


def projected_mll(P, X, Y, return_parts=False):
    """ProjectedLMCmll.forward (:1178-1241): per-datapoint MLL (scalar)."""
    n = X.shape[0]
    ytil = project_data(P, Y)                                            # :1197
    lp = gm.exact_latent_log_prob(P["kind"], X, lengthscale(P), projected_noise(P), ytil,
                                  outputscale(P), P["nu"])               # :1200-1201
    latent_res = lp.sum() / n                                            # :1202
    terms, const = projection_terms(P, Y)
    res = latent_res + sum(terms) + const                                # :1238-1240
    if return_parts:
        return res, lp, terms
    return res


def dense_lmc_log_density(P, X, Y):
    """Reference-independent identity: log N(vec(Y); 0, sum_i K_i (x) h_i h_i^T
    + I_n (x) Sigma) with Sigma = full_noise_covariance; equals n * projected_mll."""
    n, p = Y.shape
    q = P["n_latents"]
    K = gm.kernel_matrix(P["kind"], X, X, lengthscale(P), outputscale(P), P["nu"])    # (q,n,n)
    Ht = lmc_coefficients(P)                                                       # (q,p)
    Sigma = full_noise_covariance(P)
    C = torch.kron(torch.eye(n, dtype=Y.dtype), Sigma)
    for i in range(q):
        C = C + torch.kron(K[i], torch.outer(Ht[i], Ht[i]))
    return gm.mvn_log_prob(C, Y.reshape(-1))


def task_posterior(P, X, Y, Xs):
    """ProjectedGPModel.__call__ in eval mode (:1133-1155).
    Returns (mean (ns,p), covar (ns*p, ns*p) data-major interleaved, incl. +eps jitter :1153)."""
    ytil = project_data(P, Y)
    mu_lat, cov_lat = gm.exact_gp_posterior(P["kind"], X, lengthscale(P), projected_noise(P), ytil, Xs,
                                            outputscale(P), P["nu"])
    Ht = lmc_coefficients(P)                                              # (q,p)
    mean = mu_lat.T @ Ht                                                  # :1143-1146
    ns = Xs.shape[0]
    p = P["n_tasks"]
    cov = torch.zeros(ns * p, ns * p, dtype=Y.dtype)
    for i in range(P["n_latents"]):
        cov = cov + torch.kron(cov_lat[i], torch.outer(Ht[i], Ht[i]))    # :1150-1152
    cov = cov + P["eps"] * torch.eye(ns * p, dtype=Y.dtype)               # :1153
    return mean, cov


def observed_posterior(P, X, Y, Xs):
    """`full_likelihood(model(X_test))` (experiments.py:321): adds I (x) (L L^T) with
    L = jittered Cholesky factor (:1068).  Returns (mean, variance (ns,p))."""
    mean, cov = task_posterior(P, X, Y, Xs)
    L = full_noise_factor(P)
    S = L @ L.T
    var = torch.diagonal(cov).reshape(Xs.shape[0], -1) + torch.diagonal(S)[None, :]
    return mean, var
