"""Per-kernel time of k_kinv_grad at the metric shape (dev aid): python tools/kinv_time.py [n] [q]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, d, generator=g) - 1).to(dev)
y = torch.randn(q, n, generator=g).to(dev)
ell = torch.full((q, d), 0.7, device=dev, requires_grad=True)
noise = torch.full((q,), 0.7, device=dev, requires_grad=True)
for it in range(6):
    if it == 2:
        _hip.prof_enable(["k_kinv_grad"]); _hip.prof_collect()
    lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
    lp.sum().backward()
torch.cuda.synchronize()
r = _hip.prof_collect()["k_kinv_grad"]
print("k_kinv_grad n=%d q=%d: %.3f ms/launch, %.1f TF" % (n, q, r["ms"] / r["launches"], r["flops"] / r["ms"] / 1e9))
