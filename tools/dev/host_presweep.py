"""dev: host clock (ms since the step began, empty queue) at which the factorisation call of a training step is entered and left --
the host work in front of the sweep is what stands between the pivot check of one step and the next sweep (one latent per rank)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import bench
import projectedlmc as plmc
from projectedlmc import _engine
n, d, p = 8192, 8, 16
marks = {}
orig = _engine.factorize
def timed(*a, **k):
    marks["enter"] = time.perf_counter()
    r = orig(*a, **k)
    marks["leave"] = time.perf_counter()
    return r
_engine.factorize = timed
for q in [int(a) for a in sys.argv[1:]] or [1]:
    X, Y = bench.make_data(n, d, p, q, seed=0)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = plmc.ProjectedGPModel(X, Y, p, q, proj_likelihood=None, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                      init_lmc_coeffs=True, BDN=True, diagonal_B=True, scalar_B=True)
    dev = torch.device("cuda:0")
    model = model.to(dev); Xd, Yd = X.to(dev), Y.to(dev)
    model.train(); model.likelihood.train()
    mll = plmc.ProjectedLMCmll(model.likelihood, model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    def step():
        opt.zero_grad(); loss = -mll(model(Xd), Yd); loss.backward(); opt.step()
    for _ in range(5): step()
    best = None
    for _ in range(8):
        torch.cuda.synchronize(); a = time.perf_counter()
        opt.zero_grad(); out = model(Xd); b = time.perf_counter(); loss = -mll(out, Yd); c = time.perf_counter(); loss.backward(); opt.step()
        torch.cuda.synchronize()
        row = [1e3 * (x - a) for x in (b, marks["enter"], marks["leave"], c)]
        best = row if best is None or row[3] < best[3] else best
    print("q=%d  model(X) done %.2f | factorize entered %.2f, left %.2f | mll done %.2f  (ms, host clock)" % ((q,) + tuple(best)))
