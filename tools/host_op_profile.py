"""Dev aid: torch.profiler view of one training step of the bench model (which aten ops launch which device kernels,
with shapes) -- finds host-side torch work that is not negligible beside the HIP kernels."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
import projectedlmc as plmc

n, d, p, q = 8192, 8, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
