"""Test helper: load tests/golden/projected_v1.json and rebuild the product models it describes."""
import json
import os
import warnings

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PATH = os.path.join(GOLDEN, "projected_v1.json")
PATHS = [PATH, os.path.join(GOLDEN, "projected_nonbulk_v1.json")]      # round 4: bulk=False cases in a file of their own


def cases(kind):
    return [c for path in PATHS for c in json.load(open(path))["cases"] if c["kind"] == kind]


def T(x):
    return torch.tensor(x, dtype=torch.float64)


def build_projected(plmc, c):
    X, Y = T(c["X"]), T(c["Y"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = plmc.ProjectedGPModel(X, Y, c["n_tasks"], c["n_latents"], mean_type=plmc.ZeroMean,
                                  kernel_type=getattr(plmc, c["kernel"]), init_lmc_coeffs=True,
                                  outputscales=c["outputscales"], **c["ctor"]).double()
    named = dict(m.named_parameters())
    assert set(named) == set(c["params"]), (sorted(named), sorted(c["params"]))
    with torch.no_grad():
        for k, v in c["params"].items():
            named[k].copy_(T(v).reshape(named[k].shape))
        bufs = dict(m.named_buffers())
        for k, v in c.get("buffers", {}).items():
            bufs[k].copy_(T(v).reshape(bufs[k].shape))
    return m, X, Y
