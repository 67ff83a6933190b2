"""Quick timing of the exact latent log-prob + gradient at the metric shape (dev aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, settings

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
d = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, d, generator=g) - 1).to(dev, dt)
y = torch.randn(q, n, generator=g).to(dev, dt)
ell = torch.full((q, d), 0.7, dtype=dt, device=dev, requires_grad=True)
noise = torch.full((q,), 0.7, dtype=dt, device=dev, requires_grad=True)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
    torch.cuda.synchronize(); t1 = time.time()
    lp.sum().backward()
    torch.cuda.synchronize(); t2 = time.time()
    print("iter %d fwd(+grad) %.1f ms  bwd %.1f ms  logp[0]=%.6e" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), float(lp[0])), flush=True)
# size-independent check (Euler identity of the Gaussian log-density): s2 * dlogp/ds2 + sum_k ... is not available without
# an output scale, so use  y . dlogp/dy = -quad  and  logp = -(quad + logdet + n log 2 pi) / 2  consistency across two solves
yg = y.clone().requires_grad_()
lp2 = _engine.exact_latent_log_prob("matern52", X, ell.detach(), None, noise.detach(), yg)
lp2.sum().backward()
quad = -(yg.grad * y).sum(-1)
lp_half = _engine.exact_latent_log_prob("matern52", X, ell.detach(), None, noise.detach(), 0.5 * y)
print("Euler/linearity check: max rel dev %.2e" % float(((lp_half - lp2.detach()) / (0.375 * quad) - 1).abs().max()))
flop = q * n ** 3
print("n=%d q=%d %s: %.2f TFLOP/s on F_step=q*n^3" % (n, q, dt, flop / (t1 - t0) / 1e12))
