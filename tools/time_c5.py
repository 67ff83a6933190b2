"""BASELINE config 5, one rank's share (SARCOS scale: n = 44484, d = 21, one latent per GPU, Matern-5/2, fp32): time of
the exact latent log-prob + analytic gradient (the engine call of one training step), with the per-kernel HIP-event
breakdown.  Synthetic inputs of that shape."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, _hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44484
d = int(sys.argv[2]) if len(sys.argv) > 2 else 21
q = 1
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, d, generator=g) - 1).to(dev)
y = torch.randn(q, n, generator=g).to(dev)
ell = torch.full((q, d), 1.5, device=dev, requires_grad=True)
noise = torch.tensor([0.5], device=dev, requires_grad=True)
def step():
    ell.grad = None; noise.grad = None
    lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
    (-lp.sum() / n).backward()
    return lp
step(); torch.cuda.synchronize()
_hip.prof_enable(True); _hip.prof_collect()
K = 2
t0 = time.perf_counter()
for _ in range(K): lp = step()
torch.cuda.synchronize(); t1 = time.perf_counter()
st = _hip.prof_collect(); _hip.prof_enable(False)
print(json.dumps({"config": "C5 share: exact GP n=%d d=%d q=1 Matern-5/2 fp32" % (n, d), "ms_per_step": 1e3 * (t1 - t0) / K,
                  "tflops_on_n^3": n ** 3 / ((t1 - t0) / K) / 1e12, "logp": float(lp[0]),
                  "kernels": {k: {"ms": v["ms"] / K, "launches": v["launches"] / K, "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] and v["ms"] else None}
                              for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}}))
