"""The default arithmetic of the fp32 path, pinned at the METRIC SHAPE (BASELINE config 3: n = 8192, d = 8, Matern-5/2) against the
fp64 CPU oracle inside the -m gpu suite.

The bulk fp32 products run on the 16-bit matrix cores from operands split into two fp16 planes (PLMC_SPLIT=2, the default; 22
significand bits, csrc/bf3_engine.hpp) -- narrower operand arithmetic than the reference's IEEE fp32 (gpytorch on torch fp32:
projected_lmc.py:1200-1201, experiments.py:270).  This file holds what that default rests on:

 (a) one latent GP of the metric shape: log-prob and its gradient w.r.t. EVERY input (d lengthscales, the noise, all n projected
     targets) against the fp64 oracle for PLMC_SPLIT = 2, 3 and 0 (what bench.py's accuracy_checks do outside pytest);
 (b) the reference's own edge: noise AT the floor exp(-9) of projected_lmc.py:921, output scale 1, long RBF lengthscales at
     n = 8192 -- cond(Khat) eps_fp32 > 1.  The split path and the fp32-MFMA path must come off the same rung (+-1) of the jitter
     ladder with errors within 2 x of each other;
 (c) a deliberately too large eigenvalue bound must surface as info != 0 / a non-finite result, never as a finite wrong value.

The oracle runs at full size in fp64 on the host cores (oracle/cpu_step.py, ~10-30 s per evaluation)."""
import math
import os
import sys
import warnings

import pytest
import torch

from oracle import cpu_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOISE_FLOOR = math.exp(-9.0)                   # GreaterThan(exp(noise_thresh)), projected_lmc.py:920-921


@pytest.fixture(scope="module")
def eng():
    from projectedlmc import _engine
    assert torch.cuda.is_available()
    return _engine


def _host_threads():
    """bench.py's workload generator.  (No torch.set_num_threads here: changing the thread count of a process that has already
    run scikit-learn / OpenMP work -- any earlier test's SVD initialisation -- corrupted later CPU results in the same pytest
    process on the GPU box and hung on the build container; the oracle runs on torch's default thread pool.)"""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _gpu_latent(eng, kind, X, ell, noise, y):
    """log-prob + gradient vector (d + 1 + n numbers) of one latent on the HIP path, fp32; jitter warnings are returned."""
    f = lambda t: t.to(DEV, torch.float32)
    e, z, yt = f(ell)[None].requires_grad_(), f(noise).reshape(1).requires_grad_(), f(y)[None].requires_grad_()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lp = eng.exact_latent_log_prob(kind, f(X), e, None, z, yt)
        lp.sum().backward()
        torch.cuda.synchronize()
    jit = [float(str(w.message).split("jitter of ")[1].split(" ")[0]) for w in rec if "added jitter" in str(w.message)]
    g = torch.cat([e.grad.reshape(-1), z.grad.reshape(-1), yt.grad.reshape(-1)]).double().cpu()
    return float(lp), g, jit


def _oracle(kind, X, ell, noise, y):
    lp, ge, gn, gy = cpu_step.latent_step(kind, X.double(), ell.double(), noise.double(), y.double(), nu=2.5)
    return float(lp), torch.cat([ge.reshape(-1), gn.reshape(-1), gy.reshape(-1)])


@pytest.fixture(scope="module")
def metric_problem():
    """Latent 0 of the bench workload at its initial parameters (bench.make_data, SVD-initialised PLMC_fast model: SURVEY.md 8d):
    the projected targets, lengthscales ln 2 and noise softplus(0) + exp(-9) of the metric's first step."""
    bench = _host_threads()
    import projectedlmc as plmc
    n, d, p, q = 8192, 8, 16, 8
    X, Y = bench.make_data(n, d, p, q, seed=0, dtype=torch.float32)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = plmc.ProjectedGPModel(X, Y, p, q, proj_likelihood=None, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel,
                                      init_lmc_coeffs=True, BDN=True, diagonal_B=True, scalar_B=True)
    with torch.no_grad():
        ell = model.covar_module.lengthscale.reshape(q, d)[0].clone()
        noise = model.projected_noise()[0].clone()
        ytil = model.project_data(Y)[0].clone()
    ref = _oracle("matern", X, ell, noise, ytil)
    return X, ell, noise, ytil, ref


@pytest.mark.parametrize("split", ["2", "3", "0"])
def test_metric_shape_latent_against_fp64_oracle(eng, metric_problem, split):
    """(a): n = 8192, d = 8, Matern-5/2, one latent.  North-star tolerance: log-lik within 1e-4 relative; asserted here an order
    of magnitude inside what round 3 measured (1.3e-7 / 4e-7 split, 2.4e-7 / 7.8e-7 fp32 MFMA) so that a regression of the
    arithmetic shows: log-lik 5e-6, gradient (2-norm over all d + 1 + n entries) 2e-5, largest entry error 2e-5 of the largest."""
    from projectedlmc import _hip
    X, ell, noise, ytil, (lp_ref, g_ref) = metric_problem
    with _hip.knob("PLMC_SPLIT", split):
        lp, g, jit = _gpu_latent(eng, "matern52", X, ell, noise, ytil)
    assert not jit, jit
    rel = abs(lp - lp_ref) / abs(lp_ref)
    grel = float((g - g_ref).norm() / g_ref.norm())
    gmax = float((g - g_ref).abs().max() / g_ref.abs().max())
    print("PLMC_SPLIT=%s: log-lik rel err %.2e, gradient rel err %.2e (2-norm) / %.2e (max entry)" % (split, rel, grel, gmax))
    assert rel < 1e-4                                     # BASELINE.json / north_star
    assert rel < 5e-6 and grel < 2e-5 and gmax < 2e-5, (split, rel, grel, gmax)


@pytest.fixture(scope="module")
def floor_problem():
    """(b): the reference's edge.  RBF, lengthscales 2.5 on U(-1, 1)^8 (the kernel matrix is numerically rank deficient: its
    eigenvalues fall below fp32 resolution after a few hundred), output scale 1, noise exactly at the floor exp(-9) = 1.23e-4
    that GreaterThan(exp(noise_thresh)) allows (projected_lmc.py:920-921): lambda_max ~ n, cond(Khat) ~ 5e7 > 1 / eps_fp32."""
    _host_threads()
    n, d = 8192, 8
    g = torch.Generator().manual_seed(4)
    X = (2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1).float()
    ell = torch.full((d,), 2.5)
    noise = torch.tensor(NOISE_FLOOR, dtype=torch.float32)
    # targets with the structure of the model: a smooth draw (random Fourier features of the RBF spectral density) + floor noise
    om = torch.randn(512, d, generator=g, dtype=torch.float64) / 2.5
    ph = 2 * math.pi * torch.rand(512, generator=g, dtype=torch.float64)
    w = torch.randn(512, generator=g, dtype=torch.float64)
    y = math.sqrt(2.0 / 512) * (torch.cos(X.double() @ om.T + ph) @ w) + math.sqrt(NOISE_FLOOR) * torch.randn(n, generator=g, dtype=torch.float64)
    return X, ell, noise, y.float()


def test_noise_floor_split_and_fp32_paths_leave_the_jitter_ladder_together(eng, floor_problem):
    from projectedlmc import _hip, settings
    X, ell, noise, y = floor_problem
    res = {}
    for split in ("2", "0", "3"):
        # the reference's training loop runs under gp.settings.cholesky_max_tries(8) (experiments.py:265): rungs 1e-6 ... 1e-1 in fp32
        with _hip.knob("PLMC_SPLIT", split), settings.cholesky_max_tries(8):
            res[split] = _gpu_latent(eng, "rbf", X, ell, noise, y)
    rung = {s: len(r[2]) for s, r in res.items()}
    assert abs(rung["2"] - rung["0"]) <= 1 and abs(rung["3"] - rung["0"]) <= 1, rung
    # oracle at the noise each path actually factorised (floor + its last jitter; gpytorch's psd_safe_cholesky ladder)
    memo, err = {}, {}
    for s, (lp, g, jit) in res.items():
        nz = float(noise) + (jit[-1] if jit else 0.0)
        if nz not in memo:
            memo[nz] = _oracle("rbf", X, ell, torch.tensor(nz, dtype=torch.float64), y)
        lp_ref, g_ref = memo[nz]
        assert math.isfinite(lp) and torch.isfinite(g).all(), s
        err[s] = (abs(lp - lp_ref) / abs(lp_ref), float((g - g_ref).norm() / g_ref.norm()))
        print("PLMC_SPLIT=%s at the noise floor: rung %d (jitter %s), log-lik rel err %.2e, gradient rel err %.2e"
              % (s, rung[s], jit[-1] if jit else 0.0, err[s][0], err[s][1]))
    for s in ("2", "3"):
        if rung[s] == rung["0"]:                           # same matrix factorised: errors within 2 x of the fp32-MFMA path's
            assert err[s][0] <= 2.0 * err["0"][0] + 1e-6, (s, err)
            assert err[s][1] <= 2.0 * err["0"][1] + 1e-5, (s, err)
        assert err[s][0] < 1e-4, (s, err)                  # and the north-star tolerance at the reference's own edge


def _sweep(X, ell, noise, y, eig_lo):
    """plmc_assemble + plmc_potrf_ex through the C ABI with a caller-chosen eigenvalue bound; -> (logdet, quad, info)."""
    from projectedlmc import _hip, _engine
    L = _hip.lib()
    n, d = X.shape
    dt = torch.float32
    f = lambda t: t.to(DEV, dt).contiguous()
    ws = _engine.Workspace(n, 1, 1, dt, torch.device(DEV), with_inverse=True)
    st = _hip.stream_ptr(torch.device(DEV))
    Xd, ed, nd, yd = f(X), f(ell)[None].contiguous(), f(noise).reshape(1), f(y).reshape(1, 1, n)
    L.call("plmc_assemble", dt, _hip.KIND["rbf"], _hip.ptr(Xd), n, d, _hip.ptr(ed), None, _hip.ptr(nd), _hip.ptr(ws.A), ws.lda, ws.strideA, 1, st)
    L.call("plmc_write_rhs", dt, _hip.ptr(yd), 1, n, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad, 1, st)
    lo = torch.tensor([eig_lo], dtype=dt, device=DEV)
    L.call("plmc_potrf_ex", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.naug, ws.strideA, _hip.ptr(ws.Vd), _hip.ptr(ws.logdet), _hip.ptr(ws.info), 1, 1,
           _hip.ptr(lo), st)
    L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z), _hip.ptr(ws.quad), 1, st)
    # alpha = W^T z = Khat^-1 y: reads every entry of the inverse factor (the family whose scale eig_lo sets)
    L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z), _hip.ptr(ws.alpha), 1, st)
    torch.cuda.synchronize()
    return float(ws.logdet), float(ws.quad), int(ws.info), ws.alpha[0, :n].double().cpu()


def test_violated_eigenvalue_bound_never_yields_a_finite_wrong_value(floor_problem):
    """(c): the two-plane fp16 split scales the inverse-factor family by 2^13 sqrt(eig_lo) (k_split_scales).  A caller that
    overstates the bound (here by 1e2 ... 1e10 at a well-conditioned noise level, where |W| really reaches 1 / sqrt(noise))
    overflows fp16 in the planes: that must come back as info != 0 or a non-finite log det / quadratic form -- or, where the
    overstated bound still happens to cover the data, as the right number.  Never finite and wrong."""
    from projectedlmc import _hip
    X, ell, _, y = floor_problem
    n = 2304                                               # three groups of block rows: look-ahead + every split kernel
    X, y = X[:n], y[:n]
    noise = torch.tensor(1e-2)
    with _hip.knob("PLMC_SPLIT", "2"):
        ld0, q0, info0, a0 = _sweep(X, ell, noise, y, float(noise))
        assert info0 == 0 and math.isfinite(ld0) and math.isfinite(q0) and torch.isfinite(a0).all()
        with _hip.knob("PLMC_SPLIT", "0"):
            ld32, q32, _, a32 = _sweep(X, ell, noise, y, float(noise))
        close = lambda ld, qd, a: (abs(ld - ld32) < 1e-4 * abs(ld32) and abs(qd - q32) < 1e-3 * abs(q32)
                                   and float((a - a32).norm()) < 1e-2 * float(a32.norm()))
        assert close(ld0, q0, a0)
        flagged = 0
        for factor in (1e2, 1e4, 1e6, 1e8, 1e10):
            ld, qd, info, a = _sweep(X, ell, noise, y, float(noise) * factor)
            finite = info == 0 and math.isfinite(ld) and math.isfinite(qd) and bool(torch.isfinite(a).all())
            if finite:                                     # finite => right
                assert close(ld, qd, a), (factor, ld, ld32, qd, q32, float((a - a32).norm() / a32.norm()))
            else:
                flagged += 1
            print("eig_lo overstated by %.0e: info %d, log det %r, quad %r, alpha finite %s" % (factor, info, ld, qd, bool(torch.isfinite(a).all())))
        assert flagged >= 2                                # the grossly overstated bounds did overflow and were reported
