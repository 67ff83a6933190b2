"""Dev aid: host-side enqueue time vs GPU time of the exact latent log-prob at the metric shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, settings
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, 8, generator=g) - 1).to(dev)
y = torch.randn(q, n, generator=g).to(dev)
ell = torch.full((q, 8), 0.7, device=dev, requires_grad=True)
noise = torch.full((q,), 0.7, device=dev, requires_grad=True)
for chk in (True, False):
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if chk:
            lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
        else:
            with settings.check_cholesky(False):
                lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("check_cholesky=%s: host returned after %.2f ms, GPU done after %.2f ms" % (chk, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
# GPU-side span by events on torch's stream
with settings.check_cholesky(False):
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0.record()
        lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("event span %.2f ms; host enqueue %.2f ms; wall %.2f ms" % (e0.elapsed_time(e1), 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    # back-to-back calls: does the GPU pipeline them?
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(10):
        lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("10 back-to-back calls: %.2f ms per call" % (1e2 * (t2 - t0)))
