"""dev: host clock at the phase boundaries of one training step started on an empty queue (ms since the step began), and the GPU's
finish time -- where does the host wait?   python tools/dev/host_phases.py [q ...]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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

    def step(marks=None):
        t = time.perf_counter
        a = t()
        opt.zero_grad(); b0 = t()
        out = model(Xd); b1 = t()
        loss = -mll(out, Yd); b2 = t()
        loss.backward(); b3 = t()
        opt.step(); b4 = t()
        if marks is not None:
            marks.append([1e3 * (x - a) for x in (b0, b1, b2, b3, b4)])
    for _ in range(5): step()
    rows = []
    for _ in range(6):
        torch.cuda.synchronize()
        a = time.perf_counter()
        step(rows)
        torch.cuda.synchronize()
        rows[-1].append(1e3 * (time.perf_counter() - a))
    best = min(rows, key=lambda r: r[-1])
    print("q=%d  zero_grad %.2f | model(X) %.2f | mll %.2f | backward %.2f | opt.step %.2f | GPU done %.2f   (ms since the step began, "
          "host clock; PLMC_LATE_CHECK=%s)" % ((q,) + tuple(best) + (os.environ.get("PLMC_LATE_CHECK", "1"),)))
