"""BASELINE config 2: MultitaskGPModel LMC, n=2048, 8 tasks, 4 latents, RBF, fp64 -- timing of the
MLL + backward step on 1 GPU, with the per-kernel HIP-event breakdown."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import projectedlmc as plmc
from projectedlmc import _hip
n, d, p, q = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 8, 8, 4
dt = torch.float64 if (len(sys.argv) <= 2 or sys.argv[2] == "f64") else torch.float32
torch.set_default_dtype(dt)
torch.manual_seed(0)
g = torch.Generator().manual_seed(0)
X = 2 * torch.rand(n, d, generator=g) - 1
Y = torch.randn(n, p, generator=g)
lik = plmc.MultitaskGaussianLikelihood(num_tasks=p, rank=0)
model = plmc.MultitaskGPModel(X, Y, lik, n_tasks=p, n_latents=q, model_type="LMC", init_lmc_coeffs=True,
                              mean_type=plmc.ConstantMean, kernel_type=plmc.RBFKernel)
dev = torch.device("cuda:0")
model, lik = model.to(dev), lik.to(dev)
Xd, Yd = X.to(dev), Y.to(dev)
model.train(); lik.train()
mll = plmc.ExactMarginalLogLikelihood(lik, model)
opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
def step():
    opt.zero_grad(); loss = -mll(model(Xd), Yd); loss.backward(); opt.step(); return loss
for _ in range(2): l = step()
_hip.prof_enable(True); _hip.prof_collect()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 3
for _ in range(K): l = step()
torch.cuda.synchronize(); t1 = time.perf_counter()
st = _hip.prof_collect(); _hip.prof_enable(False)
N = n * p
print(json.dumps({"config": "C2 MultitaskGPModel LMC n=%d p=%d q=%d RBF %s (N=%d)" % (n, p, q, str(dt), N),
                  "ms_per_step": 1e3 * (t1 - t0) / K, "iters_per_sec": K / (t1 - t0), "loss": float(l),
                  "tflops_on_N^3": N ** 3 / ((t1 - t0) / K) / 1e12,
                  "kernels": {k: {"ms": v["ms"] / K, "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else None} for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])}}))
