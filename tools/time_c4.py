"""BASELINE config 4: VariationalMultitaskGPModel, 16 tasks, 2000 inducing points (n = 3000,
train_ind_ratio 1.5), q = 8, CholeskyVariationalDistribution, fp32 -- ELBO + backward step timing."""
import os, sys, time, json, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import projectedlmc as plmc
from projectedlmc import _hip
n, d, p, q = 3000, 8, 16, 8
g = torch.Generator().manual_seed(0)
X = 2 * torch.rand(n, d, generator=g) - 1
Y = torch.randn(n, p, generator=g)
lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=p)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = plmc.VariationalMultitaskGPModel(X, n_latents=q, n_tasks=p, train_ind_ratio=1.5, seed=0, init_lmc_coeffs=True,
                                             train_y=Y, mean_type=plmc.ZeroMean, kernel_type=plmc.RBFKernel)
dev = torch.device("cuda:0")
model, lik = model.to(dev), lik.to(dev)
Xd, Yd = X.to(dev), Y.to(dev)
mll = plmc.VariationalELBO(lik, model, num_data=n)
opt = torch.optim.AdamW([{"params": model.parameters()}, {"params": lik.parameters()}], lr=1e-2)
def step():
    opt.zero_grad(); loss = -mll(model(Xd), Yd); loss.backward(); opt.step(); return loss
for _ in range(3): l = step()
_hip.prof_enable(True); _hip.prof_collect()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): l = step()
torch.cuda.synchronize(); t1 = time.perf_counter()
st = _hip.prof_collect(); _hip.prof_enable(False)
print(json.dumps({"config": "C4 VariationalMultitaskGPModel n=%d m=%d p=%d q=%d RBF fp32" % (n, n * 2 // 3, p, q),
                  "ms_per_step": 1e3 * (t1 - t0) / K, "iters_per_sec": K / (t1 - t0), "loss": float(l),
                  "hip_kernels_ms": {k: v["ms"] / K for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])}}))
