"""Dev aid: host (CPU) time per training step vs GPU time per step of the bench model at q latents: the CPU enqueues a
step in `host` ms and the GPU executes it in `total` ms; when host >= total the step is launch-bound."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import bench
import projectedlmc as plmc

n, d, p = 8192, 8, 16
for q in [int(a) for a in sys.argv[1:]] or [1, 8]:
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
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # one isolated step: host time with an empty queue
    hs = []
    for _ in range(5):
        torch.cuda.synchronize(); a = time.perf_counter(); step(); hs.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    print("q=%d host %.2f ms/step (enqueue of %d back-to-back steps), total %.2f ms/step, isolated-step host %.2f ms" %
          (q, 1e3 * (t1 - t0) / K, K, 1e3 * (t2 - t0) / K, 1e3 * min(hs)))
