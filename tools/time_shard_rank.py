"""Dev aid: what ONE rank of the N-GPU sharded run executes per step (the full q-latent model with latent_shard =
(0, N): the projection of all q latents + the sweep of its own q / N), timed on one GPU without the all-reduce."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import bench
import projectedlmc as plmc
n, d, p, q = 8192, 8, 16, 8
for N in [int(a) for a in sys.argv[1:]] or [8, 4, 2, 1]:
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
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("rank 0 of %d (q_local = %d): %.2f ms/step" % (N, q // N if N > 1 else q, 1e3 * (t1 - t0) / K), flush=True)
