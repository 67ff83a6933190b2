#!/usr/bin/env python3
"""End-to-end example: the synthetic multi-model study of the reference's `experiments.py`
(data generation :137-170, model zoo :183-216, training loop :259-284, prediction and metrics
:288-331, :89-115), re-created on the MI355X-native `projectedlmc` package.  Nothing here is on the
measured hot path; it shows the drop-in API driving all five model kinds on one GPU.

    python examples/synthetic_study.py --n 500 --p 20 --q 5 --iters 300
"""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "projected-lmc_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import projectedlmc as plmc  # noqa: E402


def matern52(x1, x2, ell):
    r = (x1[:, None] - x2[None, :]).abs() / ell
    s = 5 ** 0.5 * r
    return (1 + s + 5.0 / 3.0 * r * r) * torch.exp(-s)


def make_data(n, n_test, p, q, q_noise, mu_noise=0.1, mu_str=0.9, min_scale=0.01, max_scale=0.5, seed=0):
    """Latent Matern GPs mixed by a random H, plus structured and unstructured noise."""
    g = torch.Generator().manual_seed(seed)
    X = torch.cat([torch.linspace(-1, 1, n, dtype=torch.float64), 2 * torch.rand(n_test, generator=g, dtype=torch.float64) - 1])
    lsc = torch.linspace(min_scale, max_scale, q, dtype=torch.float64)
    gp_vals = []
    for i in range(q):
        K = matern52(X, X, lsc[i]) + 1e-8 * torch.eye(len(X), dtype=torch.float64)
        gp_vals.append(torch.linalg.cholesky(K) @ torch.randn(len(X), generator=g, dtype=torch.float64))
    G = torch.stack(gp_vals)                                              # (q, n + n_test)
    H_true = torch.randn(q, p, generator=g, dtype=torch.float64)
    Y_sig = G.T @ H_true * (1 - mu_noise)
    H_hid = torch.randn(q_noise, p, generator=g, dtype=torch.float64)
    Y_com = torch.randn(len(X), q_noise, generator=g, dtype=torch.float64) @ H_hid * mu_str
    lev = torch.rand(p, generator=g, dtype=torch.float64) + 0.1
    Y_spec = torch.randn(len(X), p, generator=g, dtype=torch.float64) * lev.sqrt()[None, :] * (1 - mu_str)
    Y = Y_sig + (Y_com + Y_spec) * mu_noise
    X = X[:, None]
    return X[:n], Y[:n], X[n:], Y[n:]


def metrics(y_test, y_pred, sigma_pred):
    err = (y_test - y_pred).abs()
    return {"RMSE": float((err ** 2).mean().sqrt()),
            "R2": float((1 - (err ** 2).mean(0) / y_test.var(0)).mean()),
            "PVA": float(torch.log((err ** 2 / sigma_pred ** 2).mean(0)).mean()),
            "alpha_CI": float((err < 2 * sigma_pred).float().mean())}


def build(name, X, Y, p, q, kernel, mean):
    if name == "ICM" or name == "LMC":
        lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0)
        model = plmc.MultitaskGPModel(X, Y, lik, n_tasks=p, n_latents=q, model_type=name, init_lmc_coeffs=True,
                                      mean_type=mean, kernel_type=kernel)
        return model, lik, plmc.ExactMarginalLogLikelihood(lik, model), list(model.parameters())
    if name == "var":
        lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0)
        model = plmc.VariationalMultitaskGPModel(X, train_y=Y, n_tasks=p, n_latents=q, init_lmc_coeffs=True,
                                                 mean_type=mean, kernel_type=kernel, train_ind_ratio=1.5, seed=0)
        return model, lik, plmc.VariationalELBO(lik, model, num_data=X.shape[0]), list(model.parameters()) + list(lik.parameters())
    kw = {"PLMC": dict(BDN=False), "oilmm": dict(BDN=True, diagonal_B=True, scalar_B=True, diagonal_R=True),
          "PLMC_fast": dict(BDN=True, diagonal_B=True, scalar_B=True)}[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = plmc.ProjectedGPModel(X, Y, p, q, proj_likelihood=None, mean_type=mean, kernel_type=kernel,
                                      init_lmc_coeffs=True, **kw)
    return model, model.likelihood, plmc.ProjectedLMCmll(model.likelihood, model), list(model.parameters())


def run(args):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    np.random.seed(0)
    X, Y, Xt, Yt = make_data(args.n, args.n_test, args.p, args.q, args.q_noise)
    X, Y, Xt, Yt = (t.float() for t in (X, Y, Xt, Yt))
    results = {}
    for name in args.models.split(","):
        model, lik, mll, params = build(name, X, Y, args.p, args.q, plmc.MaternKernel, plmc.ZeroMean)
        model, lik = model.to(dev), lik.to(dev)
        Xd, Yd = X.to(dev), Y.to(dev)
        model.train(); lik.train()
        params = list({id(p_): p_ for p_ in (list(model.parameters()) + list(lik.parameters()))}.values())
        opt = torch.optim.AdamW(params, lr=args.lr)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda i: max(0.1, 1 - i / max(1, args.iters)))
        t0 = time.time()
        first = last = None
        for i in range(args.iters):
            opt.zero_grad()
            with plmc.settings.cholesky_max_tries(8):
                loss = -mll(model(Xd), Yd)
                loss = loss.sum()
                loss.backward()
                opt.step()
            sched.step()
            last = float(loss)
            first = last if first is None else first
        torch.cuda.synchronize()
        train_time = time.time() - t0
        model.eval(); lik.eval()
        with torch.no_grad():
            full_lik = model.full_likelihood() if hasattr(model, "full_likelihood") else lik
            pred = full_lik(model(Xt.to(dev)))
            y_pred, sigma = pred.mean.cpu(), pred.variance.clamp_min(1e-12).sqrt().cpu()
        res = metrics(Yt, y_pred, sigma)
        res.update(first_loss=first, last_loss=last, train_time=train_time, it_per_s=args.iters / train_time)
        results[name] = res
        print("%-10s" % name, json.dumps({k: round(v, 4) for k, v in res.items()}), flush=True)
    return results


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--n-test", dest="n_test", type=int, default=1000)
    ap.add_argument("--p", type=int, default=20)
    ap.add_argument("--q", type=int, default=5)
    ap.add_argument("--q-noise", dest="q_noise", type=int, default=5)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--models", default="ICM,LMC,PLMC,oilmm,var,PLMC_fast")
    run(ap.parse_args())
