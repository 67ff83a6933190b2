"""dev: is the jitter-ladder step bit-reproducible within one mode of settings.late_pivot_check, and across the two?"""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import conftest  # noqa: F401  (path set-up)
import test_gpu_projected as T
import projectedlmc as plmc
from projectedlmc import settings


def step(late):
    m, Xd, Yd = T._singular_model(plmc)
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    for prm in m.parameters():
        prm.grad = torch.full_like(prm, 0.25)
    with settings.late_pivot_check(late), warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        out = mll(m(Xd), Yd)
        (-out).backward()
    torch.cuda.synchronize()
    return float(out.detach()), {n: p.grad.clone() for n, p in m.named_parameters()}


runs = [("late", step(True)), ("early", step(False))]
os.environ["PLMC_DEFER_CHECK"] = "0"
runs.append(("direct", step(False)))
del os.environ["PLMC_DEFER_CHECK"]
for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        (a, (la, ga)), (b, (lb, gb)) = runs[i], runs[j]
        worst = max((float((ga[k] - gb[k]).abs().max()), k) for k in ga)
        print("%s#%d vs %s#%d: loss diff %.3e, worst gradient diff %.3e (%s)" % (a, i, b, j, abs(la - lb), worst[0], worst[1]))
for k, v in runs[0][1][1].items():
    print("%-60s |g - 0.25| max %.3e" % (k, float((v - 0.25).abs().max())))
for name, (l, g) in runs:
    print(name, "loss", l, {k: (float(v.flatten()[0]), float(v.abs().max())) for k, v in g.items()})
