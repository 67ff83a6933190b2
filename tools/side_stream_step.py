"""Dev aid: the training step of the bench model with torch's current stream = the default stream vs a side stream
(users that train under torch.cuda.stream(s)): ms/step and the host-side API time by call."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import bench
import projectedlmc as plmc
n, d, p, q = 8192, 8, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 1
X, Y = bench.make_data(n, d, p, q, seed=0)
dev = torch.device("cuda:0")
for side in (False, True):
    if side:
        torch.cuda.set_stream(torch.cuda.Stream(dev))
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = plmc.ProjectedGPModel(X, Y, p, q, proj_likelihood=None, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                      init_lmc_coeffs=True, BDN=True, diagonal_B=True, scalar_B=True)
    model = model.to(dev); Xd, Yd = X.to(dev), Y.to(dev)
    model.train(); model.likelihood.train()
    mll = plmc.ProjectedLMCmll(model.likelihood, model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    def step():
        opt.zero_grad(); loss = -mll(model(Xd), Yd); loss.backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("current stream = %s: %.2f ms/step" % ("side stream" if side else "default stream", 1e3 * (t1 - t0) / 20), flush=True)
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(3): step()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)[:8]
    for e in rows:
        print("    %-40s self CPU %8.2f ms  calls %d" % (e.key[:40], e.self_cpu_time_total / 1e3, e.count))
