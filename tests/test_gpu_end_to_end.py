"""End-to-end: the re-created synthetic study (examples/synthetic_study.py; loop shape of the
reference's experiments.py:259-331) trains every model kind on the GPU, the loss goes down and the
predictions are finite with sensible coverage."""
import argparse
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_synthetic_study_all_models(repo_root):
    spec = importlib.util.spec_from_file_location("synthetic_study", os.path.join(repo_root, "examples", "synthetic_study.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = argparse.Namespace(n=200, n_test=150, p=6, q=2, q_noise=2, iters=60, lr=1e-2,
                              models="ICM,LMC,PLMC,oilmm,var,PLMC_fast")
    res = mod.run(args)
    assert set(res) == {"ICM", "LMC", "PLMC", "oilmm", "var", "PLMC_fast"}
    for name, r in res.items():
        assert r["last_loss"] < r["first_loss"], (name, r)
        assert all(map(lambda v: v == v and abs(v) < 1e6, r.values())), (name, r)
        assert 0.5 < r["alpha_CI"] <= 1.0, (name, r)
