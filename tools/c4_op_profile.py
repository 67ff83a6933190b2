"""Dev aid: torch.profiler view of one ELBO + backward step of BASELINE config 4 (variational path): which ops carry the
device time and the host time outside the HIP library."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import projectedlmc as plmc
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
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=55))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=55))
