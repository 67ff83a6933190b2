import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch, warnings
from projectedlmc import _engine, _hip
dev = torch.device("cuda:0")
def run(n, d, q, dt, seed=9):
    g = torch.Generator().manual_seed(seed)
    X = (2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1).to(dev, dt)
    y = torch.randn(q, n, generator=g, dtype=torch.float64).to(dev, dt)
    ell = torch.full((q, d), 0.6931, dtype=dt, device=dev)
    noise = torch.full((q,), 0.6932, dtype=dt, device=dev)
    ws = _engine.get_workspace(n, q, 1, dt, dev, False)
    _engine.factorize("matern52", X, ell, None, noise, y.reshape(q, 1, n), ws)
    torch.cuda.synchronize()
    print(n, d, q, dt, "info", ws.info.tolist(), "logdet", ws.logdet.tolist()[:4])
    # reference: dense assemble + torch cholesky
    K = _engine.dense_cross("matern52", X, X, ell, None) + noise[:, None, None] * torch.eye(n, device=dev, dtype=dt)
    L, inf2 = torch.linalg.cholesky_ex(K)
    print("   torch info", inf2.tolist(), "logdet", (2 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)).tolist()[:4])
for cfg in [(1024, 8, 4, torch.float32), (1024, 8, 2, torch.float32), (1024, 8, 4, torch.float64), (1000, 8, 4, torch.float32), (2048, 8, 4, torch.float32), (8192, 8, 8, torch.float32)]:
    run(*cfg)
