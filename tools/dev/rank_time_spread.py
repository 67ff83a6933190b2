"""dev: per-block step times of one rank's shard (blocks of 5 steps, no sync inside a block) -- is a slow run one hiccup or a slower step?
   python tools/dev/rank_time_spread.py N [N ...]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import bench
import projectedlmc as plmc
n, d, p, q = 8192, 8, 16, 8
for N in [int(a) for a in sys.argv[1:]] or [4]:
    X, Y = bench.make_data(n, d, p, q, seed=0)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = plmc.ProjectedGPModel(X, Y, p, q, proj_likelihood=None, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                      init_lmc_coeffs=True, BDN=True, diagonal_B=True, scalar_B=True,
                                      latent_shard=(0, N) if N > 1 else None)
    dev = torch.device("cuda:0")
    model = model.to(dev); Xd, Yd = X.to(dev), Y.to(dev)
    model.train(); model.likelihood.train()
    mll = plmc.ProjectedLMCmll(model.likelihood, model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    def step():
        opt.zero_grad(); loss = -mll(model(Xd), Yd); loss.backward(); opt.step()
    for _ in range(5): step()
    out = []
    for b in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / 5)
    print("N=%d (q_local=%d): blocks of 5 steps, ms/step: %s" % (N, q // N, " ".join("%.2f" % x for x in out)), flush=True)
