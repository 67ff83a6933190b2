"""Dev aid: A/B of library knobs in ONE process, interleaved rounds (cdna_hip_programming.md rule 24).
    python tools/ab_knobs.py Q "NAME=VAL,NAME=VAL" "NAME=VAL" ...      (each argument = one variant; "" = defaults)
Times the exact latent log-prob + gradient (assembly, sweep, K^-1 + gradient kernel) at n = 8192, d = 8, q = Q, fp32 and
prints per variant the median / min ms over the rounds and the median sweep time (HIP-event bracket of plmc_potrf)."""
import contextlib, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, _hip

q = int(sys.argv[1])
variants = sys.argv[2:] or [""]
n, d = int(os.environ.get("AB_N", "8192")), 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, d, generator=g) - 1).to(dev)
y = torch.randn(q, n, generator=g).to(dev)
ell = torch.full((q, d), 0.7, device=dev, requires_grad=True)
noise = torch.full((q,), 0.7, device=dev, requires_grad=True)


def once():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lp = _engine.exact_latent_log_prob("matern52", X, ell, None, noise, y)
    lp.sum().backward()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), float(lp[0])


def ctx(v):
    st = contextlib.ExitStack()
    for kv in filter(None, v.split(",")):
        k, val = kv.split("=")
        st.enter_context(_hip.knob(k, val))
    return st


res = {v: [] for v in variants}
sw = {v: [] for v in variants}
vals = {}
rounds = int(os.environ.get("AB_ROUNDS", "6"))
for r in range(rounds + 1):
    for v in variants:
        with ctx(v):
            once()                                  # settle (knob change: first call may allocate / re-plan)
            _hip.prof_enable(("sweep_total",)); _hip.prof_collect()
            t, val = once()
            s = _hip.prof_collect().get("sweep_total", {"ms": float("nan")})["ms"]
            _hip.prof_enable(False)
        if r:                                        # round 0 = warm-up
            res[v].append(t); sw[v].append(s); vals[v] = val
for v in variants:
    print("%-50s step %7.2f ms (min %7.2f)  sweep %7.2f ms   logp[0] %.6e" % (v or "(defaults)", statistics.median(res[v]), min(res[v]),
                                                                           statistics.median(sw[v]), vals[v]))
