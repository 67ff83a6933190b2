"""Dev aid: how long does the HOST take to queue one sweep (plmc_potrf_ex: ~250 launches + ~100 event calls on three streams)
against how long the GPU takes to run it?  If the host is not well ahead, the chain of a single-latent shard idles on it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, _hip
n, d = 8192, 8
dev = torch.device("cuda:0")
for q in (1, 2, 8):
    g = torch.Generator().manual_seed(0)
    X = (2 * torch.rand(n, d, generator=g) - 1).to(dev)
    y = torch.randn(q, 1, n, generator=g).to(dev)
    ell = torch.full((q, d), 0.7, device=dev)
    noise = torch.full((q,), 0.7, device=dev)
    ws = _engine.Workspace(n, q, 1, torch.float32, dev, True)
    host, total = [], []
    for r in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _engine.factorize("matern52", X, ell, None, noise, y, ws)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if r:
            host.append(1e3 * (t1 - t0)); total.append(1e3 * (t2 - t0))
    print("q = %d: host queues assemble + sweep in %.2f ms (min %.2f); done on the GPU after %.2f ms" % (q, sorted(host)[len(host) // 2], min(host), sorted(total)[len(total) // 2]))
    del ws
    torch.cuda.empty_cache()
