"""Dev aid: per-kernel-class times of the exact latent log-prob + gradient: python tools/kernel_times.py [n] [q] [f64]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
d = 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, d, generator=g) - 1).to(dev, dt)
y = torch.randn(q, n, generator=g).to(dev, dt)
ell = torch.full((q, d), 0.7, device=dev, dtype=dt, requires_grad=True)
noise = torch.full((q,), 0.7, device=dev, dtype=dt, requires_grad=True)
for it in range(6):
    if it == 2:
        _hip.prof_enable(True); _hip.prof_collect()
    lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
    lp.sum().backward()
torch.cuda.synchronize()
for k, r in _hip.prof_collect().items():
    print("%-16s %8.3f ms/step  %4d launches/step  %7.1f TF  %7.1f GB/s" % (
        k, r["ms"] / 4, r["launches"] // 4, r["flops"] / max(r["ms"], 1e-9) / 1e9, r["bytes"] / max(r["ms"], 1e-9) / 1e6))
print("logp[0] = %.6f" % float(lp[0]))
