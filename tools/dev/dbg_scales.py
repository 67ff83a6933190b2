import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd"), os.path.join(ROOT, "tests")]
import torch
from projectedlmc import _engine as eng, _hip
from oracle import gp_math as gm
from test_gpu_engine import _problem
n, d = 2300, 6
dev = torch.device("cuda:0")
f = lambda t: t.to(dev, torch.float32)
for seed, nz, osv, ysc in ((1, 2e-3, 1.0, 1.0), (2, 80.0, 4e4, 300.0), (3, 2e-6, 1e-3, 1e-2), (4, 0.3, 1.0, 1e4)):
    X, y, ell, noise, osc = _problem(n, d, 2, seed=seed)
    noise = torch.full_like(noise, nz); osc = torch.full_like(osc, osv); y = y * ysc
    ref = gm.exact_latent_log_prob_analytic("matern", X, ell, noise, y, osc, 2.5)
    for split in ("2", "3", "0"):
        with _hip.knob("PLMC_SPLIT", split):
            ell_d, nz_d, y_d, os_d = f(ell).requires_grad_(), f(noise).requires_grad_(), f(y).requires_grad_(), f(osc).requires_grad_()
            lp = eng.exact_latent_log_prob("matern52", f(X), ell_d, os_d, nz_d, y_d)
            lp.sum().backward(); torch.cuda.synchronize()
            errs = [float(((lp.detach().cpu().double() - ref[0]).abs() / ref[0].abs()).max())]
            for got, want in ((ell_d.grad, ref[1]), (nz_d.grad, ref[2]), (os_d.grad, ref[3]), (y_d.grad, ref[4])):
                errs.append(float((got.detach().cpu().double() - want).abs().max() / want.abs().max()))
            print("case", seed, "split", split, " rel errs lp/ell/noise/os/y:", " ".join("%.2e" % e for e in errs), flush=True)
