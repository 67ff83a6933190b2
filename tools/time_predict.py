"""Timing of the eval-mode prediction path: ProjectedGPModel(X*) at the metric shape (C3: n = 8192, d = 8,
p = 16, q = 8, n* = 2048) and one rank's share of BASELINE config 5 (n = 44 484, d = 21, q = 1 latent of p = 7 tasks,
n* = 4 449), fp32.  First call = augmented sweep (factorisation kept: _engine.PosteriorCache); later calls = cross assembly +
forward substitution of the new columns (plmc_potrs_aug) + posterior moments + task mixing.
    python tools/time_predict.py [c3|c5] -> one JSON line; kernel classes from the in-library HIP-event profiler."""
import json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
import projectedlmc as plmc
from projectedlmc import _hip

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n, d, p, q, ns, shard = (8192, 8, 16, 8, 2048, None) if cfg == "c3" else (44484, 21, 7, 7, 4449, (0, 7))
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = 2 * torch.rand(n, d, generator=g) - 1
Y = torch.randn(n, p, generator=g)
Xs = (2 * torch.rand(ns, d, generator=g) - 1).to(dev)
torch.manual_seed(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    kw = dict(latent_shard=shard) if shard else {}
    m = plmc.ProjectedGPModel(X, Y, p, q, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel, init_lmc_coeffs=True, BDN=True, scalar_B=True,
                              diagonal_B=True, **kw)
if shard:                                   # one rank's share: the all-reduce of the partial sums is a no-op stand-in here
    from projectedlmc import parallel
    parallel.all_reduce_sum = lambda t: t
m = m.to(dev).eval()
q_loc = q if not shard else len(range(shard[0], q, shard[1]))


def call():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        out = m(Xs)
        mean, var = out.mean, out.variance
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), float(mean.abs().max()), float(var.min())

first, mx, vmin = call()                      # plain augmented sweep (the lazy cache only remembers the state)
build, _, _ = call()                          # second call with unchanged state: the sweep with the inverse factor + kept planes
_hip.prof_enable(True); _hip.prof_collect()
times = [call()[0] for _ in range(5)]
prof = _hip.prof_collect(); _hip.prof_enable(False)
c = m._prediction_cache()
n_pad = c.ws.n_pad
algo_bytes = q_loc * n_pad * (1 + ns) * 4.0          # posterior moments: one read of the augmented columns
res = {"config": cfg, "n": n, "d": d, "n_star": ns, "latents_on_this_rank": q_loc, "dtype": "f32",
       "first_call_ms": first, "cache_build_call_ms": build, "cached_call_ms": sorted(times)[len(times) // 2], "cache": {"hits": c.hits, "misses": c.misses},
       "kernels_per_cached_call": {k: {"ms": v["ms"] / 5, "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] > 0 else None,
                                        "gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["bytes"] > 0 else None}
                                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
       "posterior_moments_algorithmic_bytes": algo_bytes, "sanity": {"max_abs_mean": mx, "min_var": vmin}}
print(json.dumps(res))
