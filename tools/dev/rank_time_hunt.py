"""dev: the q_local = 2 shard sometimes steps in 7.8-8.1 ms instead of 5.9 (time_shard_rank sequences 8 4 2 1).  Repeat the sequence;
when a (model instance, N) pair is slow, print the per-kernel table (HIP-event brackets) and buffer addresses."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import bench
import projectedlmc as plmc
from projectedlmc import _hip, _engine
import gc
GC_LOG = []
def _gc_cb(phase, info):
    if phase == "start":
        GC_LOG.append([time.perf_counter(), info["generation"], None])
    else:
        GC_LOG[-1][2] = time.perf_counter() - GC_LOG[-1][0]
        GC_LOG[-1].append(info.get("collected", 0))
gc.callbacks.append(_gc_cb)
def mem():
    st = torch.cuda.memory_stats()
    return st.get("num_device_alloc", 0), st.get("num_device_free", 0), st.get("num_alloc_retries", 0)
n, d, p, q = 8192, 8, 16, 8
for rep in range(6):
    for N in (8, 4, 2, 1):
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
        marks = []
        del GC_LOG[:]
        m0 = mem()
        for _ in range(K):
            step(); marks.append(time.perf_counter())
        m1 = mem()
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / K
        per = [1e3 * (b - a) for a, b in zip([t0] + marks[:-1], marks)]
        tag = ""
        if N == 4:
            _hip.prof_enable(True); _hip.prof_collect()
            for _ in range(3): step()
            torch.cuda.synchronize()
            t = _hip.prof_collect(); _hip.prof_enable(False)
            tag = "  " + " ".join("%s %.2f" % (k, v["ms"] / 3) for k, v in sorted(t.items(), key=lambda kv: -kv[1]["ms"])[:7])
            ws = [w for w in getattr(_engine, "_WS_CACHE", {}).values()] if hasattr(_engine, "_WS_CACHE") else []
            tag += "  A@%s" % ",".join(hex(w.A.data_ptr()) for w in ws[-1:]) if ws else ""
        print("rep %d N=%d (q_local=%d): %.2f ms/step  host per step min %.2f max %.2f (step %d)%s%s" % (rep, N, q // N, ms, min(per), max(per), per.index(max(per)),
              "  ALL: " + " ".join("%.1f" % x for x in per) + "  device alloc/free/retries %s -> %s  gc: %s" % (m0, m1, [(round(1e3 * (g[0] - t0), 1), g[1], round(1e3 * (g[2] or 0), 1), g[3:]) for g in GC_LOG]) if max(per) > 1.5 * min(per) + 1 else "", ""), flush=True)
        del model, mll, opt
