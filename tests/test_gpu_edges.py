"""Edge cases of the HIP path: tiny / ragged / block-aligned n, d = 1 and d = 32 (plmc_max_dim),
many latents, p == q, odd numbers of 128-blocks (exercises the pair/look-ahead schedule), repeated
calls with cached workspaces, and size-independent properties at a larger size."""
import warnings

import pytest
import torch

from oracle import gp_math as gm
from oracle import projected as pj
from _bridge import oracle_params, perturb_

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(eng, kind, okind, nu, n, d, q, seed, dtype=torch.float64, use_os=True):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    y = torch.randn(q, n, generator=g, dtype=torch.float64)
    ell = 0.4 + 0.6 * torch.rand(q, d, generator=g, dtype=torch.float64)
    if d > 8:
        ell = ell * (d / 4) ** 0.5
    noise = 0.05 + 0.5 * torch.rand(q, generator=g, dtype=torch.float64)
    osc = 0.5 + torch.rand(q, generator=g, dtype=torch.float64) if use_os else None
    ref = gm.exact_latent_log_prob_analytic(okind, X, ell, noise, y, osc, nu)
    f = lambda t: None if t is None else t.to(DEV, dtype)
    ell_d, nz_d, y_d = f(ell).requires_grad_(), f(noise).requires_grad_(), f(y).requires_grad_()
    lp = eng.exact_latent_log_prob(kind, f(X), ell_d, f(osc), nz_d, y_d)
    lp.sum().backward()
    return lp.detach().cpu().double(), ell_d.grad.cpu().double(), nz_d.grad.cpu().double(), y_d.grad.cpu().double(), ref


@pytest.fixture(scope="module")
def eng():
    from projectedlmc import _engine
    assert torch.cuda.is_available()
    return _engine


@pytest.mark.parametrize("n", [1, 2, 17, 127, 128, 129, 255, 256, 257, 383, 384, 640, 769])
def test_ragged_and_aligned_sizes(eng, n):
    lp, ge, gn, gy, ref = _run(eng, "matern52", "matern", 2.5, n, 3, 2, seed=n)
    assert torch.allclose(lp, ref[0], rtol=1e-10)
    assert torch.allclose(ge, ref[1], rtol=1e-7, atol=1e-9)
    assert torch.allclose(gn, ref[2], rtol=1e-7, atol=1e-9)
    assert torch.allclose(gy, ref[4], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("d", [1, 5, 9, 16, 21, 32])
def test_input_dimensions(eng, d):
    lp, ge, gn, gy, ref = _run(eng, "rbf", "rbf", 2.5, 300, d, 2, seed=d)
    assert torch.allclose(lp, ref[0], rtol=1e-10)
    assert torch.allclose(ge, ref[1], rtol=1e-7, atol=1e-9)


def test_dimension_above_limit_raises(eng):
    X = torch.zeros(10, 33, device=DEV, dtype=torch.float64)
    with pytest.raises(ValueError, match="plmc_max_dim"):
        eng.exact_latent_log_prob("rbf", X, torch.ones(1, 33, device=DEV, dtype=torch.float64), None,
                                  torch.ones(1, device=DEV, dtype=torch.float64),
                                  torch.zeros(1, 10, device=DEV, dtype=torch.float64))


def test_many_latents_and_repeated_calls(eng):
    for rep in range(3):                                   # cached workspace must give identical results
        lp, ge, gn, gy, ref = _run(eng, "matern32", "matern", 1.5, 200, 2, 11, seed=5)
        assert torch.allclose(lp, ref[0], rtol=1e-10)
        assert torch.allclose(ge, ref[1], rtol=1e-7, atol=1e-9)


def test_p_equals_q_projected_model():
    import projectedlmc as plmc
    g = torch.Generator().manual_seed(3)
    n, d, p = 150, 2, 3
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    Y = torch.randn(n, p, generator=g, dtype=torch.float64)
    for kw in (dict(BDN=True, scalar_B=True, diagonal_B=True), dict(BDN=False)):
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = plmc.ProjectedGPModel(X, Y, p, p, mean_type=plmc.ZeroMean, kernel_type=plmc.RBFKernel,
                                      init_lmc_coeffs=True, **kw)
        m = perturb_(m.double())
        P = oracle_params(m)
        ref = float(pj.projected_mll(P, X, Y))
        m = m.to(DEV)
        m.train()
        val = float(plmc.ProjectedLMCmll(m.likelihood, m)(m(X.to(DEV)), Y.to(DEV)))
        assert abs(val - ref) < 1e-9 * abs(ref), (kw, val, ref)


def test_full_size_properties_fp32(eng):
    """At n = 4096 (too big for the dense fp64 oracle to be quick) check size-independent properties:
    log-prob invariance under a permutation of the data points, linear scaling identity
    logp(c y; c^2 K) = logp(y; K) - n log c, and d logp/dy = -alpha with alpha^T y = quadratic form."""
    n, d, q = 4096, 8, 2
    g = torch.Generator().manual_seed(0)
    X = (2 * torch.rand(n, d, generator=g) - 1).to(DEV)
    y = torch.randn(q, n, generator=g).to(DEV)
    ell = torch.full((q, d), 0.8, device=DEV)
    noise = torch.tensor([0.3, 0.6], device=DEV)
    osc = torch.tensor([1.0, 1.0], device=DEV)
    yg = y.clone().requires_grad_()
    lp = eng.exact_latent_log_prob("matern52", X, ell, osc, noise, yg)
    lp.sum().backward()
    perm = torch.randperm(n, generator=g).to(DEV)
    lp_perm = eng.exact_latent_log_prob("matern52", X[perm], ell, osc, noise, y[:, perm])
    assert torch.allclose(lp.detach(), lp_perm, rtol=2e-5)
    c = 3.0
    lp_scaled = eng.exact_latent_log_prob("matern52", X, ell, osc * c * c, noise * c * c, y * c)
    assert torch.allclose(lp_scaled, lp.detach() - n * torch.log(torch.tensor(c)), rtol=2e-5)
    # Euler identity for the Gaussian log-density: y . dlogp/dy = -quad, and logp = -1/2(quad + logdet + n log 2pi)
    quad = -(yg.grad * y).sum(-1)
    assert bool((quad > 0).all())
    lp_half = eng.exact_latent_log_prob("matern52", X, ell, osc, noise, 0.5 * y)
    assert torch.allclose(lp_half - lp.detach(), 0.5 * quad * (1 - 0.25), rtol=1e-3)


def test_metric_shape_properties_fp32(eng):
    """The BASELINE.json metric shape itself (n = 8192, d = 8, q = 8 latents, Matern-5/2, fp32), where the dense
    fp64 oracle is out of reach of a test: identities that tie the separately computed outputs together.
      (1) solve round trip: Khat alpha = y with alpha = -dlogp/dy, Khat rebuilt by the cross-assembly kernel;
      (2) scaling (Euler) identity of the gradient kernel: os dlogp/dos + s2 dlogp/ds2 = (quad - n) / 2;
      (3) trace link between the LOO path and the gradient path: dlogp/ds2 = (alpha.alpha - tr Khat^-1) / 2 with
          tr Khat^-1 = sum_i 1 / sigma2_loo_i;
      (4) permutation invariance of the log-density."""
    n, d, q = 8192, 8, 8
    g = torch.Generator().manual_seed(3)
    X = (2 * torch.rand(n, d, generator=g) - 1).to(DEV)
    y = torch.randn(q, n, generator=g).to(DEV)
    ell = torch.linspace(0.5, 1.2, q)[:, None].expand(q, d).contiguous().to(DEV)
    noise = torch.linspace(0.2, 0.9, q).to(DEV)
    osc = torch.linspace(0.8, 1.5, q).to(DEV)
    yg, ng, og = y.clone().requires_grad_(), noise.clone().requires_grad_(), osc.clone().requires_grad_()
    lp = eng.exact_latent_log_prob("matern52", X, ell, og, ng, yg)
    lp.sum().backward()
    alpha = -yg.grad
    quad = (alpha * y).sum(-1)
    assert bool((quad > 0).all())
    # (1) residual of Khat alpha = y, latent by latent (one 8192 x 8192 fp32 matrix at a time)
    for i in range(q):
        K = eng.dense_cross("matern52", X, X, ell[i:i + 1], osc[i:i + 1])[0]
        r = K @ alpha[i] + noise[i] * alpha[i] - y[i]
        assert float(r.norm() / y[i].norm()) < 2e-4, i
        del K
    # (2) Euler identity
    lhs = osc * og.grad + noise * ng.grad
    assert torch.allclose(lhs, 0.5 * (quad - n), rtol=2e-4, atol=1e-2)
    # (3) trace link with the LOO outputs
    s2, _ = eng.exact_loo("matern52", X, ell, osc, noise, y)
    tr_kinv = (1.0 / s2.double()).sum(-1)
    assert torch.allclose(ng.grad.double(), 0.5 * ((alpha.double() ** 2).sum(-1) - tr_kinv), rtol=2e-4, atol=1e-2)
    # (4) permutation invariance
    perm = torch.randperm(n, generator=g).to(DEV)
    lp_perm = eng.exact_latent_log_prob("matern52", X[perm], ell, osc, noise, y[:, perm])
    assert torch.allclose(lp.detach(), lp_perm, rtol=2e-5)


def test_sarcos_scale_single_latent_fp32(eng):
    """One rank's share of BASELINE config 5 (n = 44484, d = 21, one latent per GPU, Matern-5/2, fp32; 16 GB factor
    buffer): the engine runs at that size (64-bit offsets, 348 block rows) and its outputs -- the d = 21 instance of the
    fused K^-1 + gradient kernel included -- satisfy the identities that tie them together:
      (1) linearity of the Gaussian log-density: logp(y/2) - logp(y) = 3/8 * y.(Khat^-1 y);
      (2) scaling (Euler) identity of the gradient kernel: os dlogp/dos + s2 dlogp/ds2 = (quad - n) / 2;
      (3) trace link with the LOO path: dlogp/ds2 = (alpha.alpha - tr Khat^-1) / 2, tr Khat^-1 = sum_i 1 / sigma2_loo_i;
      (4) the 21 lengthscale gradients against a central difference of the log-density along the direction ell
          (d/dc logp(c ell) at c = 1 = sum_k ell_k dlogp/dell_k)."""
    n, d, q = 44484, 21, 1
    g = torch.Generator().manual_seed(0)
    X = (2 * torch.rand(n, d, generator=g) - 1).to(DEV)
    y = torch.randn(q, n, generator=g).to(DEV)
    ell = (1.2 + 0.6 * torch.rand(q, d, generator=g)).to(DEV)
    noise = torch.tensor([0.5], device=DEV)
    osc = torch.tensor([1.3], device=DEV)
    yg, eg, ng, og = y.clone().requires_grad_(), ell.clone().requires_grad_(), noise.clone().requires_grad_(), osc.clone().requires_grad_()
    lp = eng.exact_latent_log_prob("matern52", X, eg, og, ng, yg)
    lp.sum().backward()
    assert bool(torch.isfinite(lp).all()) and bool(torch.isfinite(eg.grad).all())
    alpha = -yg.grad
    quad = (alpha * y).sum(-1)
    assert bool((quad > 0).all())
    # (1)
    lp_half = eng.exact_latent_log_prob("matern52", X, ell, osc, noise, 0.5 * y)
    assert torch.allclose(lp_half - lp.detach(), 0.375 * quad, rtol=1e-4)
    # (2)
    lhs = osc * og.grad + noise * ng.grad
    assert torch.allclose(lhs, 0.5 * (quad - n), rtol=2e-4, atol=5e-2), (lhs, 0.5 * (quad - n))
    # (3)
    s2, _ = eng.exact_loo("matern52", X, ell, osc, noise, y)
    tr_kinv = (1.0 / s2.double()).sum(-1)
    assert torch.allclose(ng.grad.double(), 0.5 * ((alpha.double() ** 2).sum(-1) - tr_kinv), rtol=2e-4, atol=5e-2)
    # (4) fp32 log-densities of size ~1e5 carry ~1e-2 of rounding: a 2 % step keeps the difference well above it
    h = 0.02
    lp_p = eng.exact_latent_log_prob("matern52", X, ell * (1 + h), osc, noise, y)
    lp_m = eng.exact_latent_log_prob("matern52", X, ell * (1 - h), osc, noise, y)
    fd = (lp_p - lp_m).double() / (2 * h)
    an = (ell.double() * eg.grad.double()).sum(-1)
    assert torch.allclose(fd, an, rtol=2e-2, atol=1.0), (fd, an)
    eng.free_workspaces()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,q,dtype,split", [(8192, 8, torch.float32, 2), (8192, 2, torch.float32, 2), (4096, 4, torch.float64, 2),
                                               (8192, 4, torch.float32, 3), (8192, 4, torch.float32, 0), (8192, 8, torch.float32, 0)],
                         ids=["metric-f32-q8-wsplit", "f32-q2-two-streams", "f64-q4", "f32-q4-bf16x3", "f32-q4-fp32-mfma", "metric-f32-q8-fp32-mfma"])
def test_sweep_is_deterministic_and_schedule_independent(eng, n, q, dtype, split):
    """The look-ahead runs the chain, the head and the tail updates of the sweep on two or three streams (three with
    q >= 4: the inverse-factor columns get their own chain); every tile still receives its updates in a fixed order, so
    (a) repeated factorisations must agree bit for bit over the WHOLE factor buffer (U, augmented column, W), and
    (b) they must agree bit for bit with the one-stream schedule (PLMC_SERIAL=1).  A race between the streams, or a
    store-data hazard in the tile write-back ("Write-back hazard": round 1's "wrong factors under the
    look-ahead" was exactly that and showed up here as thousands of differing tiles), fails this test.
    With the default arithmetic of the fp32 path (bulk products as the two-plane fp16 split on the matrix cores,
    bf3_engine.hpp: which tiles take which engine does not depend on the schedule), with the three-plane bf16 split
    (PLMC_SPLIT=3) and with PLMC_SPLIT=0 (fp32 MFMA everywhere)."""
    import os
    import contextlib
    from projectedlmc import _hip
    with (contextlib.nullcontext() if split == 2 else _hip.knob("PLMC_SPLIT", str(split))):
        _schedule_independence_body(eng, n, q, dtype)
        eng.free_workspaces()


def _schedule_independence_body(eng, n, q, dtype):
    import os
    d = 8
    g = torch.Generator().manual_seed(5)
    X = (2 * torch.rand(n, d, generator=g, dtype=dtype) - 1).to(DEV)
    y = torch.randn(q, n, generator=g, dtype=dtype).to(DEV)
    ell = torch.linspace(0.4, 1.0, q, dtype=dtype)[:, None].expand(q, d).contiguous().to(DEV)
    noise = torch.linspace(0.05, 0.5, q, dtype=dtype).to(DEV)
    ws = eng.Workspace(n, q, 1, dtype, torch.device(DEV), True)
    it = torch.int32 if dtype == torch.float32 else torch.int64

    def factor():
        eng.factorize("matern52", X, ell, None, noise, y.reshape(q, 1, n), ws)
        torch.cuda.synchronize()
        return ws.A.view(it).clone(), ws.logdet.clone(), ws.info.clone()

    from projectedlmc import _hip
    assert "PLMC_SERIAL" not in os.environ
    with _hip.knob("PLMC_SERIAL", "1"):
        ref, ld_ref, info_ref = factor()
    assert not bool(info_ref.any()) and bool(torch.isfinite(ld_ref).all())
    for rep in range(4):
        A, ld, info = factor()
        ndiff = int((A != ref).sum())
        assert ndiff == 0, "run %d: %d elements of the factor buffer differ from the one-stream schedule" % (rep, ndiff)
        assert torch.equal(ld, ld_ref) and torch.equal(info, info_ref)
        del A
    # (c) the chain of a group as ONE resident launch (k_chain: workgroups handing tiles over through version counters) performs
    # the tile operations of the three launches per block row it replaces, in the same order per tile: the launch-per-step chain
    # (PLMC_CHAIN=0) and other pool sizes (PLMC_CHAIN_NW: 1, 5, 200 workgroups beside the q critical ones) give the same bits
    for knob, val in (("PLMC_CHAIN", "0"), ("PLMC_CHAIN_NW", "1"), ("PLMC_CHAIN_NW", "5"), ("PLMC_CHAIN_NW", "200")):
        with _hip.knob(knob, val):
            A, ld, info = factor()
        ndiff = int((A != ref).sum())
        assert ndiff == 0, "%s=%s: %d elements of the factor buffer differ" % (knob, val, ndiff)
        assert torch.equal(ld, ld_ref) and torch.equal(info, info_ref)
        del A
    # (d) assembly and sweep as one library call (plmc_factorize_ex_*: only the first group's rows are assembled in front of the chain,
    # the others beside it -- the default of eng.factorize, so every run above was one) against plmc_assemble_* + plmc_potrf_ex_*
    os.environ["PLMC_FUSED_ASSEMBLE"] = "0"
    try:
        A, ld, info = factor()
    finally:
        del os.environ["PLMC_FUSED_ASSEMBLE"]
    ndiff = int((A != ref).sum())
    assert ndiff == 0, "two calls instead of plmc_factorize_ex: %d elements of the factor buffer differ" % ndiff
    assert torch.equal(ld, ld_ref) and torch.equal(info, info_ref)
    del A
    del ref, ws
    torch.cuda.empty_cache()


def test_sweeps_from_three_caller_streams_overlap_safely(eng):
    """The helper streams and ordering events of the look-ahead are bound to the CALLER stream (two sets per device, api.hip):
    sweeps queued on two streams at once (own buffers) run concurrently on their own sets, a third stream takes over the least
    recently used set and must first wait for that set's last sweep.  All three factor buffers must equal, bit for bit, what
    the same sweeps give one after the other on one stream."""
    n, q, d, dtype = 4096, 2, 6, torch.float32
    g = torch.Generator().manual_seed(11)
    X = (2 * torch.rand(n, d, generator=g, dtype=dtype) - 1).to(DEV)
    it = torch.int32
    probs = []
    for k in range(3):
        y = torch.randn(q, n, generator=g, dtype=dtype).to(DEV)
        ell = (0.4 + 0.2 * k + 0.3 * torch.rand(q, d, generator=g, dtype=dtype)).to(DEV)
        noise = (0.05 + 0.1 * k + 0.2 * torch.rand(q, generator=g, dtype=dtype)).to(DEV)
        probs.append((ell, noise, y.reshape(q, 1, n).contiguous(), eng.Workspace(n, q, 1, dtype, torch.device(DEV), True)))
    torch.cuda.synchronize()
    ref = []
    for ell, noise, y, ws in probs:                              # one after the other, one stream
        eng.factorize("matern52", X, ell, None, noise, y, ws)
        torch.cuda.synchronize()
        ref.append((ws.A.view(it).clone(), ws.logdet.clone()))
        ws.A.zero_()
    streams = [torch.cuda.Stream(DEV) for _ in probs]
    for rep in range(8):
        torch.cuda.synchronize()
        for (ell, noise, y, ws), s in zip(probs, streams):       # all three in flight together
            with torch.cuda.stream(s):
                eng.factorize("matern52", X, ell, None, noise, y, ws)
        torch.cuda.synchronize()
        for k, ((ell, noise, y, ws), (A_ref, ld_ref)) in enumerate(zip(probs, ref)):
            ndiff = int((ws.A.view(it) != A_ref).sum())
            assert ndiff == 0, "round %d, stream %d: %d elements differ from the sequential sweep" % (rep, k, ndiff)
            assert torch.equal(ws.logdet, ld_ref)
    del probs, ref
    torch.cuda.empty_cache()


def test_sweeps_from_two_host_threads_are_serialised_by_the_library(eng):
    """the library's process-global bookkeeping (sweep contexts, profiler record) was documented as single-threaded
    but not enforced.  Entry points now hold one library-wide lock while they enqueue.  Two host threads (ctypes releases the GIL
    during a call) factorise their own problems on their own streams, many times, with the profiler recording: every factor buffer
    equals the one the same problem gives alone, bit for bit."""
    import threading
    from projectedlmc import _hip
    n, q, d, dtype = 2304, 2, 5, torch.float32
    g = torch.Generator().manual_seed(21)
    X = (2 * torch.rand(n, d, generator=g, dtype=dtype) - 1).to(DEV)
    it = torch.int32
    probs = []
    for k in range(2):
        y = torch.randn(q, n, generator=g, dtype=dtype).to(DEV)
        ell = (0.5 + 0.2 * k + 0.3 * torch.rand(q, d, generator=g, dtype=dtype)).to(DEV)
        noise = (0.05 + 0.1 * k + 0.2 * torch.rand(q, generator=g, dtype=dtype)).to(DEV)
        probs.append((ell, noise, y.reshape(q, 1, n).contiguous(), eng.Workspace(n, q, 1, dtype, torch.device(DEV), True)))
    ref = []
    for ell, noise, y, ws in probs:
        eng.factorize("matern52", X, ell, None, noise, y, ws)
        torch.cuda.synchronize()
        ref.append((ws.A.view(it).clone(), ws.logdet.clone()))
    errors = []

    def worker(k):
        try:
            ell, noise, y, ws = probs[k]
            s = torch.cuda.Stream(DEV)
            for rep in range(12):
                with torch.cuda.stream(s):
                    eng.factorize("matern52", X, ell, None, noise, y, ws)
                s.synchronize()
                if int((ws.A.view(it) != ref[k][0]).sum()) != 0 or not torch.equal(ws.logdet, ref[k][1]):
                    errors.append((k, rep))
        except Exception as exc:                           # a worker must not die silently
            errors.append((k, repr(exc)))

    _hip.prof_enable(True)
    try:
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        stats = _hip.prof_collect()
    finally:
        _hip.prof_enable(False)
    assert not errors, errors
    assert stats["sweep_total"]["launches"] == 24
    del probs, ref
    torch.cuda.empty_cache()


def test_training_step_is_deterministic_at_metric_shape(eng):
    """Whole MLL + gradient evaluation (sweep, alpha, fused K^-1 + gradient kernel on its own stream) repeated at the
    metric shape: log-probs and every gradient bit-identical."""
    n, d, q = 8192, 8, 8
    g = torch.Generator().manual_seed(5)
    X = (2 * torch.rand(n, d, generator=g) - 1).to(DEV)
    y = torch.randn(q, n, generator=g).to(DEV)
    ell = torch.linspace(0.4, 1.0, q)[:, None].expand(q, d).contiguous().to(DEV)
    noise = torch.linspace(0.05, 0.5, q).to(DEV)
    outs = []
    for _ in range(4):
        yg, eg = y.clone().requires_grad_(), ell.clone().requires_grad_()
        lp = eng.exact_latent_log_prob("matern52", X, eg, None, noise, yg)
        lp.sum().backward()
        outs.append((lp.detach().clone(), yg.grad.clone(), eg.grad.clone()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
    assert bool(torch.isfinite(outs[0][0]).all())


def test_many_latents_go_through_the_chain_kernel_in_batches(eng):
    """The resident chain kernel carries at most 32 latents per launch (its critical workgroups must all be resident); q = 70 small
    problems of two groups each run as three launches per group.  Same bits as the launch-per-step chain, log-probs against a
    per-latent fp64 torch reference."""
    from projectedlmc import _hip
    n, q, d, dtype = 1300, 70, 3, torch.float64
    g = torch.Generator().manual_seed(8)
    X = (2 * torch.rand(n, d, generator=g, dtype=dtype) - 1).to(DEV)
    y = torch.randn(q, n, generator=g, dtype=dtype).to(DEV)
    ell = (0.4 + 0.6 * torch.rand(q, d, generator=g, dtype=dtype)).to(DEV)
    noise = (0.05 + 0.3 * torch.rand(q, generator=g, dtype=dtype)).to(DEV)
    ws = eng.Workspace(n, q, 1, dtype, torch.device(DEV), True)

    def factor():
        eng.factorize("rbf", X, ell, None, noise, y.reshape(q, 1, n), ws)
        torch.cuda.synchronize()
        return ws.A.view(torch.int64).clone(), ws.logdet.clone(), ws.info.clone()

    A1, ld1, info1 = factor()
    with _hip.knob("PLMC_CHAIN", "0"):
        A0, ld0, info0 = factor()
    assert not bool(info1.any()) and int((A1 != A0).sum()) == 0 and torch.equal(ld1, ld0)
    for l in (0, 31, 32, 69):
        r2 = torch.cdist(X / ell[l], X / ell[l]) ** 2
        K = torch.exp(-0.5 * r2) + noise[l] * torch.eye(n, dtype=dtype, device=DEV)
        assert abs(float(torch.logdet(K)) - float(ld1[l])) < 1e-8 * abs(float(ld1[l]))
    del ws, A0, A1
    torch.cuda.empty_cache()
