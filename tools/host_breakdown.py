"""Dev aid: host-side (Python) time of the sections of one training step at the metric shape, steady state."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import projectedlmc as plmc
from projectedlmc import _engine, _hip

q = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, d, p = 8192, 8, 16
g = torch.Generator().manual_seed(0)
X = 2 * torch.rand(n, d, generator=g) - 1
Y = torch.randn(n, p, generator=g)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = plmc.ProjectedGPModel(X, Y, p, q, proj_likelihood=None, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                  init_lmc_coeffs=True, BDN=True, diagonal_B=True, scalar_B=True)
dev = torch.device("cuda:0")
model = model.to(dev); Xd, Yd = X.to(dev), Y.to(dev)
model.train(); model.likelihood.train()
mll = plmc.ProjectedLMCmll(model.likelihood, model)
opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
T = {}
def tick(name, t0):
    t1 = time.perf_counter(); T[name] = T.get(name, 0.0) + (t1 - t0); return t1
# wrap the engine pieces
orig_factorize = _engine.factorize
def timed_factorize(*a, **k):
    t0 = time.perf_counter(); r = orig_factorize(*a, **k); tick("  engine: factorize enqueue (assemble+rhs+potrf)", t0); return r
_engine.factorize = timed_factorize
orig_failed = _engine._DeferredInfo.failed
def timed_failed(self):
    t0 = time.perf_counter(); r = orig_failed(self); tick("  engine: wait for pivot check (GPU sweep)", t0); return r
_engine._DeferredInfo.failed = timed_failed
orig_pd = model.project_data
def timed_pd(Yv):
    t0 = time.perf_counter(); r = orig_pd(Yv); tick("  mll: project_data", t0); return r
model.project_data = timed_pd
orig_elp = _engine.exact_latent_log_prob
def timed_elp(*a, **k):
    t0 = time.perf_counter(); r = orig_elp(*a, **k); tick("  mll: exact_latent_log_prob (all)", t0); return r
_engine.exact_latent_log_prob = timed_elp
K = 20
for it in range(K + 5):
    if it == 5:
        T.clear(); torch.cuda.synchronize(); tstart = time.perf_counter()
    t = time.perf_counter()
    opt.zero_grad(); t = tick("zero_grad", t)
    out = model(Xd); t = tick("model(X)", t)
    loss = -mll(out, Yd); t = tick("mll.forward (all)", t)
    loss.backward(); t = tick("backward", t)
    opt.step(); t = tick("opt.step", t)
torch.cuda.synchronize(); total = time.perf_counter() - tstart
print("q=%d: %.2f ms/step wall" % (q, 1e3 * total / K))
for k, v in T.items():
    print("%-55s %7.3f ms/step" % (k, 1e3 * v / K))
