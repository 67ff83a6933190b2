"""dev: which piece of the SGPR eval path depends on uninitialised device memory?  Poison the caching allocator's free memory with
NaN (or large finite values), then run WhitenedInterp / spd_half_solve against torch references."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _var_engine, _dense, _engine
dev = torch.device("cuda:0")
val = float(sys.argv[1]) if len(sys.argv) > 1 else float("nan")
def poison():
    _engine.free_workspaces(); _var_engine._ws.clear()
    torch.cuda.empty_cache()
    junk = [torch.full((256, 1024, 1024), val, dtype=torch.float32, device=dev) for _ in range(8)]   # 8 GB
    del junk
torch.manual_seed(0)
m, q, n, d, ns = 25, 2, 160, 2, 10
Z = (2 * torch.rand(m, d, dtype=torch.float64) - 1).to(dev)
X = (2 * torch.rand(n, d, dtype=torch.float64) - 1).to(dev)
ell = (0.3 + torch.rand(q, d, dtype=torch.float64)).to(dev)
def ref_interp(Xq):
    out = []
    for i in range(q):
        def k(a, b):
            r = torch.cdist(a / ell[i], b / ell[i]); s5 = 5 ** 0.5
            return (1 + s5 * r + 5 * r * r / 3) * torch.exp(-s5 * r)
        L = torch.linalg.cholesky(k(Z, Z))
        out.append(torch.linalg.solve_triangular(L, k(Z, Xq), upper=False))
    return torch.stack(out)
for rep in range(3):
    poison()
    with torch.no_grad():
        A = _var_engine.whitened_interp("matern52", Z, X, ell, None, 0.0)
        As = _var_engine.whitened_interp("matern52", Z, X[:ns], ell, None, 0.0)
    eA, eAs = (A - ref_interp(X)).abs().max().item(), (As - ref_interp(X[:ns])).abs().max().item()
    poison()
    g = torch.Generator().manual_seed(1)
    M0 = torch.randn(q, m, m, generator=g, dtype=torch.float64).to(dev)
    M = M0 @ M0.transpose(-1, -2) + m * torch.eye(m, dtype=torch.float64, device=dev)
    R = torch.randn(q, m, 11, generator=g, dtype=torch.float64).to(dev)
    sol = _dense.spd_half_solve(M, R)
    L = torch.linalg.cholesky(M)
    eS = (sol - torch.linalg.solve_triangular(L, R, upper=False)).abs().max().item()
    print("rep %d poison %r: interp err %.2e / %.2e, half-solve err %.2e" % (rep, val, eA, eAs, eS), flush=True)
