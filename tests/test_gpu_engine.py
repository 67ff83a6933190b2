"""GPU parity of the HIP hot path (through the C ABI) against the CPU oracle."""
import pytest
import torch

from oracle import gp_math as gm

pytestmark = pytest.mark.gpu

KINDS = {"rbf": ("rbf", 2.5), "matern12": ("matern", 0.5), "matern32": ("matern", 1.5), "matern52": ("matern", 2.5)}


def _problem(n, d, q, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    X = 2 * torch.rand(n, d, generator=g, dtype=torch.float64) - 1
    y = torch.randn(q, n, generator=g, dtype=torch.float64)
    ell = 0.3 + 0.5 * torch.rand(q, d, generator=g, dtype=torch.float64)
    noise = 0.05 + 0.5 * torch.rand(q, generator=g, dtype=torch.float64)
    osc = 0.5 + torch.rand(q, generator=g, dtype=torch.float64)
    return X, y, ell, noise, osc


@pytest.fixture(scope="module")
def eng():
    from projectedlmc import _engine
    assert torch.cuda.is_available()
    return _engine


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("n,d,q", [(128, 2, 1), (200, 3, 2), (517, 8, 3)])
@pytest.mark.parametrize("use_os", [False, True])
def test_logprob_and_grad_fp64(eng, kind, n, d, q, use_os):
    okind, nu = KINDS[kind]
    X, y, ell, noise, osc = _problem(n, d, q, seed=n + d)
    if kind == "matern12":
        pass
    ref = gm.exact_latent_log_prob_analytic(okind, X, ell, noise, y, osc if use_os else None, nu)
    dev = torch.device("cuda:0")
    ell_d = ell.to(dev).requires_grad_()
    nz_d = noise.to(dev).requires_grad_()
    y_d = y.to(dev).requires_grad_()
    os_d = osc.to(dev).requires_grad_() if use_os else None
    lp = eng.exact_latent_log_prob(kind, X.to(dev), ell_d, os_d, nz_d, y_d)
    w = torch.linspace(0.5, 1.5, q, dtype=torch.float64)
    (lp * w.to(dev)).sum().backward()
    assert torch.allclose(lp.detach().cpu(), ref[0], rtol=1e-10, atol=0), (lp.detach().cpu(), ref[0])
    assert torch.allclose(ell_d.grad.cpu(), w[:, None] * ref[1], rtol=1e-7, atol=1e-9)
    assert torch.allclose(nz_d.grad.cpu(), w * ref[2], rtol=1e-7, atol=1e-9)
    assert torch.allclose(y_d.grad.cpu(), w[:, None] * ref[4], rtol=1e-7, atol=1e-9)
    if use_os:
        assert torch.allclose(os_d.grad.cpu(), w * ref[3], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("n,d,q,env", [(1300, 3, 2, None), (2100, 4, 8, None), (3300, 3, 2, None), (1300, 3, 2, "PLMC_SERIAL=1"),
                                       (2100, 4, 8, "PLMC_GRP=4"), (1300, 3, 2, "PLMC_GRP=3"), (2100, 4, 8, "PLMC_HALF_TILES=0"),
                                       (1300, 3, 2, "PLMC_HALF_TILES=1"), (1300, 3, 2, "PLMC_GRAD_STREAM=0"),
                                       (1300, 3, 2, "PLMC_KINV_IN_SWEEP=1"), (2100, 4, 8, "PLMC_KINV_IN_SWEEP=1"),
                                       (1300, 3, 2, "PLMC_KINV_ORDER=1"), (1300, 3, 2, "PLMC_KINV_ORDER=0"),
                                       (1300, 3, 2, "PLMC_KINV_ORDER=5")])
def test_multi_group_sweep_fp64(eng, n, d, q, env, monkeypatch):
    """Sizes at which the sweep runs its look-ahead schedule (more than two groups of block rows: chain, group panel +
    head rows and tail on three streams; ragged last group), against the dense fp64 oracle; also with the look-ahead
    off, other group sizes, half / full tiles for the small launches, the gradient stream and the tile order / general
    epilogue of the gradient kernel switched (the dev knobs must not change results)."""
    from projectedlmc import _hip
    if env and env.startswith(("PLMC_GRAD_STREAM", "PLMC_KINV_IN_SWEEP")):
        monkeypatch.setenv(*env.split("="))                     # Python-level knob, read per call
    elif env:
        with _hip.knob(*env.split("=")):
            return _multi_group_body(eng, n, d, q)
    return _multi_group_body(eng, n, d, q)


def _multi_group_body(eng, n, d, q):
    X, y, ell, noise, osc = _problem(n, d, q, seed=n)
    ref = gm.exact_latent_log_prob_analytic("matern", X, ell, noise, y, osc, 2.5)
    dev = torch.device("cuda:0")
    ell_d, nz_d = ell.to(dev).requires_grad_(), noise.to(dev).requires_grad_()
    y_d, os_d = y.to(dev).requires_grad_(), osc.to(dev).requires_grad_()
    lp = eng.exact_latent_log_prob("matern52", X.to(dev), ell_d, os_d, nz_d, y_d)
    lp.sum().backward()
    assert torch.allclose(lp.detach().cpu(), ref[0], rtol=1e-10, atol=0)
    assert torch.allclose(ell_d.grad.cpu(), ref[1], rtol=1e-7, atol=1e-9)
    assert torch.allclose(nz_d.grad.cpu(), ref[2], rtol=1e-7, atol=1e-9)
    assert torch.allclose(os_d.grad.cpu(), ref[3], rtol=1e-7, atol=1e-9)
    assert torch.allclose(y_d.grad.cpu(), ref[4], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("n,d,q", [(1000, 8, 2), (2300, 12, 3), (4200, 5, 2)])
def test_split_engines_against_fp32_mfma_path(eng, n, d, q, monkeypatch):
    """Arithmetic of the bulk fp32 products (the depth-1024 trailing updates and group panels of the sweep, from three groups
    of block rows on, and the W^T W products of the gradient kernel; csrc/bf3_engine.hpp):
      default (PLMC_SPLIT=2): two fp16 planes per operand, scaled by bounds from the diagonal and the noise, three plane
                              products into two fp32 accumulator levels;
      PLMC_SPLIT=3:           three bf16 planes, six plane products, two levels;
      PLMC_SPLIT=0:           v_mfma_f32_16x16x4_f32 everywhere.
    All must sit inside the fp32 tolerance against the fp64 oracle, and the split engines' errors must be of the size of the
    fp32 MFMA path's or below (the isolated product is 0.3-0.45 x, profiles/r03_split_numerics.txt; through the
    ill-conditioned solve these are different roundings of the same problem, so they are compared through the oracle with
    a factor 2 + a few fp32 ulps of the largest magnitude, never with each other)."""
    from projectedlmc import _hip
    X, y, ell, noise, osc = _problem(n, d, q, seed=n + 1)
    ref = gm.exact_latent_log_prob_analytic("matern", X, ell, noise, y, None, 2.5)
    dev = torch.device("cuda:0")
    f = lambda t: t.to(dev, torch.float32)

    def run():
        ell_d, nz_d, y_d = f(ell).requires_grad_(), f(noise).requires_grad_(), f(y).requires_grad_()
        lp = eng.exact_latent_log_prob("matern52", f(X), ell_d, None, nz_d, y_d)
        lp.sum().backward()
        torch.cuda.synchronize()
        return [t.detach().cpu().double() for t in (lp, ell_d.grad, nz_d.grad, y_d.grad)]

    h2 = run()
    with _hip.knob("PLMC_SPLIT", "3"):
        b3 = run()
    with _hip.knob("PLMC_SPLIT", "0"):
        plain = run()
    # K^-1 accumulated group by group inside the sweep (split engine: from the planes of the group's rows of W, on a filler
    # stream) + the one-pass gradient kernel, instead of the fused K^-1 + gradient kernel behind the sweep
    monkeypatch.setenv("PLMC_KINV_IN_SWEEP", "1")
    h2_in_sweep = run()
    with _hip.knob("PLMC_SPLIT", "3"):
        b3_in_sweep = run()
    monkeypatch.delenv("PLMC_KINV_IN_SWEEP")
    for split in (h2, b3, h2_in_sweep, b3_in_sweep):
        for got, base, want, tol in ((split[0], plain[0], ref[0], 1e-4), (split[1], plain[1], ref[1], 2e-3),
                                     (split[2], plain[2], ref[2], 2e-3), (split[3], plain[3], ref[4], 2e-3)):
            scale = want.abs().max()
            e_split, e_plain = (got - want).abs().max() / scale, (base - want).abs().max() / scale
            assert e_split < tol and e_plain < tol, (e_split, e_plain)
            assert e_split < 2.0 * e_plain + 2e-6, (e_split, e_plain)


@pytest.mark.parametrize("n,q", [(900, 2), (2500, 3)])
def test_kinv_grad_from_the_sweeps_planes_equals_the_split_pass(eng, n, q):
    """plmc_kinv_grad_vd_* multiplies the planes of W the sweep left in its Vd scratch; plmc_kinv_grad_ex_* splits W again
    into its own scratch.  Same fp32 values, same power-of-two scale, same kernel: the gradients must be IDENTICAL (one group
    of block rows: inverse triangle only; three groups: group panels too).  A sweep without eigenvalue bounds (three bf16
    planes) followed by a K^-1 call with them (two fp16 planes) must not go unnoticed: the scratch records its scheme and the
    mismatch poisons the gradients."""
    from projectedlmc import _hip
    L = _hip.lib()
    d, dt, dev = 5, torch.float32, torch.device("cuda:0")
    X, y, ell, noise, osc = _problem(n, d, q, seed=n)
    f = lambda t: t.to(dev, dt).contiguous()
    Xd, yd, elld, nzd = f(X), f(y), f(ell), f(noise)
    ws = eng.Workspace(n, q, 1, dt, dev, True)
    st = _hip.stream_ptr(dev)

    def grads(entry, with_bounds_in_sweep=True, **kw):
        L.call("plmc_assemble", dt, _hip.KIND["matern52"], _hip.ptr(Xd), n, d, _hip.ptr(elld), None, _hip.ptr(nzd), _hip.ptr(ws.A), ws.lda, ws.strideA, q, st)
        L.call("plmc_write_rhs", dt, _hip.ptr(yd), 1, n, _hip.ptr(ws.A), ws.lda, ws.strideA, 0, ws.naug_pad, q, st)
        L.call("plmc_potrf_ex", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.naug, ws.strideA, _hip.ptr(ws.Vd), _hip.ptr(ws.logdet), _hip.ptr(ws.info), 1, q,
               _hip.ptr(nzd) if with_bounds_in_sweep else None, st)
        L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z), _hip.ptr(ws.quad), q, st)
        L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z), _hip.ptr(ws.alpha), q, st)
        g = torch.zeros(q, d + 2, dtype=torch.float64, device=dev)
        kd = torch.zeros(q, ws.n_pad, dtype=dt, device=dev)
        if entry == "vd":
            part = torch.empty(int(L.cdll.plmc_grad_partials_bytes(ws.n_pad, q)) // 8, dtype=torch.float64, device=dev)
            L.call("plmc_kinv_grad_vd", dt, _hip.KIND["matern52"], _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.alpha), _hip.ptr(Xd), n, d,
                   _hip.ptr(elld), None, _hip.ptr(g), None, 0, 0, _hip.ptr(kd), _hip.ptr(part), q, _hip.ptr(nzd), _hip.ptr(ws.Vd), st)
        else:
            part = torch.empty(int(L.cdll.plmc_grad_scratch_bytes_for(ws.n_pad, q, 4)) // 8, dtype=torch.float64, device=dev)
            L.call("plmc_kinv_grad_ex", dt, _hip.KIND["matern52"], _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.alpha), _hip.ptr(Xd), n, d,
                   _hip.ptr(elld), None, _hip.ptr(g), None, 0, 0, _hip.ptr(kd), _hip.ptr(part), q, _hip.ptr(nzd), st)
        torch.cuda.synchronize()
        return g.cpu(), kd.cpu()

    for knob in ("2", "3"):
        with _hip.knob("PLMC_SPLIT", knob):
            g_vd, kd_vd = grads("vd")
            g_ex, kd_ex = grads("ex")
        assert torch.isfinite(g_vd).all() and g_vd.abs().max() > 0
        assert torch.equal(g_vd, g_ex) and torch.equal(kd_vd, kd_ex), (knob, (g_vd - g_ex).abs().max())
    g_bad, _ = grads("vd", with_bounds_in_sweep=False)        # three-plane sweep, two-plane K^-1 call
    assert torch.isnan(g_bad).all()


def test_fp16_split_handles_extreme_scales(eng):
    """The two-plane fp16 split scales its operands by bounds derived from the largest diagonal entry and the noise, so that
    fp16 never overflows and small magnitudes keep their bits: tiny noise (W entries ~ 1 / sqrt(noise) large), large and
    tiny output scales (U entries ~ sqrt(outputscale)), and targets of any size (the augmented columns stay on the fp32
    MFMAs).  Same fp32 tolerance against the fp64 oracle as anywhere else."""
    n, d = 2300, 6
    dev = torch.device("cuda:0")
    f = lambda t: t.to(dev, torch.float32)
    for seed, nz, osv, ysc in ((1, 2e-3, 1.0, 1.0), (2, 80.0, 4e4, 300.0), (3, 2e-6, 1e-3, 1e-2), (4, 0.3, 1.0, 1e4)):
        X, y, ell, noise, osc = _problem(n, d, 2, seed=seed)
        noise = torch.full_like(noise, nz)
        osc = torch.full_like(osc, osv)
        y = y * ysc
        ref = gm.exact_latent_log_prob_analytic("matern", X, ell, noise, y, osc, 2.5)
        ell_d, nz_d, y_d, os_d = f(ell).requires_grad_(), f(noise).requires_grad_(), f(y).requires_grad_(), f(osc).requires_grad_()
        lp = eng.exact_latent_log_prob("matern52", f(X), ell_d, os_d, nz_d, y_d)
        lp.sum().backward()
        torch.cuda.synchronize()
        assert torch.isfinite(lp).all()
        assert ((lp.detach().cpu().double() - ref[0]).abs() / ref[0].abs()).max() < 1e-4, (seed, lp, ref[0])
        for got, want in ((ell_d.grad, ref[1]), (nz_d.grad, ref[2]), (os_d.grad, ref[3]), (y_d.grad, ref[4])):
            assert (got.detach().cpu().double() - want).abs().max() < 2e-3 * want.abs().max(), seed


@pytest.mark.parametrize("kind", ["rbf", "matern52"])
def test_logprob_and_grad_fp32(eng, kind):
    """fp32 tolerance: log-prob within 1e-4 relative of the fp64 oracle (BASELINE.json target),
    gradients within 2e-3 of their max magnitude."""
    okind, nu = KINDS[kind]
    n, d, q = 1000, 8, 2
    X, y, ell, noise, osc = _problem(n, d, q, seed=7)
    ref = gm.exact_latent_log_prob_analytic(okind, X, ell, noise, y, None, nu)
    dev = torch.device("cuda:0")
    f = lambda t: t.to(dev, torch.float32)
    ell_d = f(ell).requires_grad_()
    nz_d = f(noise).requires_grad_()
    y_d = f(y).requires_grad_()
    lp = eng.exact_latent_log_prob(kind, f(X), ell_d, None, nz_d, y_d)
    lp.sum().backward()
    rel = ((lp.detach().cpu().double() - ref[0]) / ref[0]).abs().max()
    assert rel < 1e-4, rel
    for got, want in ((ell_d.grad, ref[1]), (nz_d.grad, ref[2]), (y_d.grad, ref[4])):
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert err < 2e-3, err


def test_posterior_fp64(eng):
    n, d, q, ns = 300, 3, 2, 77
    X, y, ell, noise, osc = _problem(n, d, q, seed=3)
    Xs = 2 * torch.rand(ns, d, dtype=torch.float64) - 1
    mu, cov = gm.exact_gp_posterior("matern", X, ell, noise, y, Xs, osc, 2.5)
    dev = torch.device("cuda:0")
    m1, v1 = eng.exact_posterior("matern52", X.to(dev), ell.to(dev), osc.to(dev), noise.to(dev), y.to(dev), Xs.to(dev))
    assert torch.allclose(m1.cpu(), mu, rtol=1e-8, atol=1e-10)
    assert torch.allclose(v1.cpu(), torch.diagonal(cov, dim1=-2, dim2=-1), rtol=1e-7, atol=1e-10)
    m2, c2 = eng.exact_posterior("matern52", X.to(dev), ell.to(dev), osc.to(dev), noise.to(dev), y.to(dev), Xs.to(dev),
                                 full_cov=True)
    assert torch.allclose(c2.cpu(), cov, rtol=1e-7, atol=1e-9)


def test_not_pd_is_reported(eng):
    """A non-PD matrix must walk the jitter ladder (warning) -- duplicate points with ~zero noise."""
    n, d, q = 130, 2, 1
    X, y, ell, noise, _ = _problem(n, d, q)
    X[1] = X[0]
    dev = torch.device("cuda:0")
    nz = torch.full((1,), -1e-3, dtype=torch.float64)          # negative "noise": indefinite
    with pytest.warns(RuntimeWarning):
        with pytest.raises(RuntimeError):
            eng.exact_latent_log_prob("rbf", X.to(dev), ell.to(dev), None, nz.to(dev), y.to(dev))


def test_cpu_tensors_fail_loudly(eng):
    X, y, ell, noise, _ = _problem(64, 2, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        eng.exact_latent_log_prob("rbf", X, ell, None, noise, y)


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("d", [3, 8, 11])
def test_assembled_covariance_entries_fp32(eng, kind, d):
    """Entries of the assembled fp32 covariance (packed kernel for d <= 8 with the refined hardware exp, generic kernel
    above) against the fp64 oracle: within ~4 ulp of the largest entry (fp32 scaled inputs u = x / ell, fp32 polynomial
    times exp), distances from 0 to far in the tail."""
    from projectedlmc import _hip
    n, q = 300, 2
    X, y, ell, noise, osc = _problem(n, d, q, seed=5)
    X = X * 3.0                                     # scaled distances up to ~ 20
    dev = torch.device("cuda:0")
    ws = eng.get_workspace(n, q, 1, torch.float32, dev, False)
    L = _hip.lib()
    f = lambda t: t.to(dev, torch.float32).contiguous()
    Xc, ec, nc = f(X), f(ell), f(noise)
    L.call("plmc_assemble", torch.float32, _hip.KIND[kind], _hip.ptr(Xc), n, d, _hip.ptr(ec), None, _hip.ptr(nc),
           _hip.ptr(ws.A), ws.lda, ws.strideA, q, _hip.stream_ptr(dev))
    torch.cuda.synchronize()
    k, nu = KINDS[kind]
    for i in range(q):
        # oracle on the fp32-rounded inputs, so that only the arithmetic of the kernel is compared
        Kref = gm.kernel_matrix(k, Xc.double().cpu(), Xc.double().cpu(), ec[i].double().cpu()[None], nu=nu)[0]
        Kref = Kref + nc[i].double().cpu() * torch.eye(n, dtype=torch.float64)
        got = ws.A[i, :n, :n].double().cpu()
        err = (got - Kref).triu().abs().max()
        assert float(err) < 6e-7 * (1.0 + float(nc[i])), (kind, d, float(err))


def test_non_current_device(eng):
    """The library keys its helper streams / events on the CURRENT device: a model that lives on another device than the
    current one must still work.  Needs two GPUs; the 1-GPU test boxes skip it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    assert torch.cuda.current_device() == 0
    dev = torch.device("cuda:1")
    X, y, ell, noise, osc = _problem(700, 3, 2, seed=3)
    ref = gm.exact_latent_log_prob_analytic("matern", X, ell, noise, y, osc, 2.5)
    ell_d = ell.to(dev).requires_grad_()
    lp = eng.exact_latent_log_prob("matern52", X.to(dev), ell_d, osc.to(dev), noise.to(dev), y.to(dev))
    lp.sum().backward()
    assert torch.cuda.current_device() == 0
    assert torch.allclose(lp.detach().cpu(), ref[0], rtol=1e-10)
    assert torch.allclose(ell_d.grad.cpu(), ref[1], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_wt_matvec_column_split_is_bit_identical(eng, dt):
    """alpha = W^T z over the lower block triangle.  A launch with fewer than 128 (latent, block column) pairs uses workgroups of 32
    columns instead of 128 (four times the workgroups for a one-latent shard); every column's sum is built from the same row groups
    in the same order, so a latent's result must not depend on how many latents share the launch -- and it must be W^T z."""
    from projectedlmc import _hip
    L = _hip.lib()
    dev = torch.device("cuda:0")
    n_pad, q = 2048, 8                                        # 16 block columns: q = 8 -> 128 pairs (128-column form), q = 1 -> 16 (32-column form)
    torch.manual_seed(5)
    W = torch.randn(q, n_pad, n_pad, dtype=dt, device=dev)
    z = torch.randn(q, n_pad, dtype=dt, device=dev)
    st = _hip.stream_ptr(dev)
    batch = torch.zeros(q, n_pad, dtype=dt, device=dev)
    L.call("plmc_wt_matvec", dt, _hip.ptr(W), n_pad, n_pad, n_pad * n_pad, _hip.ptr(z), _hip.ptr(batch), q, st)
    single = torch.zeros(q, n_pad, dtype=dt, device=dev)
    for i in range(q):
        L.call("plmc_wt_matvec", dt, _hip.ptr(W[i]), n_pad, n_pad, n_pad * n_pad, _hip.ptr(z[i]), _hip.ptr(single[i]), 1, st)
    torch.cuda.synchronize()
    assert torch.equal(batch, single)
    # reference: rows l >= the first row of column i's block (what the kernel reads of a lower-triangular W stored with full blocks)
    blk = torch.arange(n_pad, device=dev) // 128
    mask = (blk[:, None] >= blk[None, :]).to(torch.float64)    # [l, i]
    want = torch.einsum("qli,ql->qi", W.double() * mask, z.double())
    tol = 1e-12 if dt == torch.float64 else 2e-6
    assert (batch.double() - want).abs().max() <= tol * want.abs().max()
