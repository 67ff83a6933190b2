"""CPU tests of the host-side mirror (no GPU, no compute calls into the HIP library):
the p x q projection algebra against the oracle, constructor semantics / error behaviour of the
reference, state-dict names, the C-ABI export list, and the loud failure without a GPU."""
import ctypes
import os
import re
import warnings

import pytest
import torch

import projectedlmc as plmc
from oracle import projected as pj
from _bridge import oracle_params, perturb_



@pytest.fixture(autouse=True)
def _float64_default():
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(old)

VARIANTS = {
    "PLMC": dict(BDN=False, diagonal_B=False, scalar_B=False),
    "PLMC_diagB": dict(BDN=False, diagonal_B=True, scalar_B=False),
    "BDN_fullB": dict(BDN=True, diagonal_B=False, scalar_B=False),
    "BDN_diagB": dict(BDN=True, diagonal_B=True, scalar_B=False),
    "PLMC_fast": dict(BDN=True, diagonal_B=True, scalar_B=True),
}


def _data(n=60, d=3, p=6, seed=0):
    g = torch.Generator().manual_seed(seed)
    return 2 * torch.rand(n, d, generator=g) - 1, torch.randn(n, p, generator=g)


def _model(X, Y, q, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return plmc.ProjectedGPModel(X, Y, Y.shape[1], q, mean_type=plmc.ZeroMean, kernel_type=plmc.MaternKernel, **kw)


@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("init", [False, True])
def test_projection_algebra_matches_oracle(name, init):
    X, Y = _data()
    torch.manual_seed(3)
    m = perturb_(_model(X, Y, 2, init_lmc_coeffs=init, **VARIANTS[name]))
    P = oracle_params(m)
    assert torch.allclose(m.project_data(Y), pj.project_data(P, Y), atol=1e-12)
    assert torch.allclose(m.projection_matrix(), pj.projection_matrix(P), atol=1e-12)
    assert torch.allclose(m.full_noise_covariance(), pj.full_noise_covariance(P), atol=1e-12)
    assert torch.allclose(m.lmc_coefficients(), pj.lmc_coefficients(P), atol=1e-14)
    assert torch.allclose(m.B_tilde(), pj.B_tilde(P), atol=1e-12)
    assert torch.allclose(m.projected_noise(), pj.projected_noise(P), atol=1e-14)
    fl = m.full_likelihood()
    assert torch.allclose(fl.task_noise_covar_factor.data, pj.full_noise_factor(P), atol=1e-10)
    assert not hasattr(fl, "noise")                      # has_global_noise=False (experiments.py:323)


@pytest.mark.parametrize("name", list(VARIANTS))
def test_initial_parameters_match_oracle_init(name):
    """Same initial values as projected_lmc.py:916-993 (SVD init path is deterministic)."""
    X, Y = _data(n=80, p=5)
    m = _model(X, Y, 3, init_lmc_coeffs=True, **VARIANTS[name])
    P0 = pj.init_params(X, Y, 3, kind="matern", init_lmc_coeffs=True, **VARIANTS[name])
    P = oracle_params(m)
    # H is defined up to the sign of each singular vector; compare the sign-invariant products
    assert torch.allclose(P["H"] @ P["H"].T, P0["H"] @ P0["H"].T, atol=1e-10)
    assert torch.allclose(pj.full_noise_covariance(P), pj.full_noise_covariance(P0), atol=1e-10)
    for k in ("raw_noise", "raw_lengthscale", "log_B_tilde", "B_tilde_inv_chol_raw", "M"):
        if k in P0:
            assert torch.allclose(P[k], P0[k], atol=1e-12), k


def test_svd_init_matches_sklearn_convention():
    """init_lmc_coefficients (projected_lmc.py:183-201): same numbers as sklearn's randomized_svd
    whenever its range finder is exact (q + 10 >= p), including the svd_flip sign convention."""
    X, Y = _data(n=100, p=7)
    for q in (2, 5, 7):
        ours = plmc.init_lmc_coefficients(Y, q)
        ref = pj.svd_init(Y, q)
        assert torch.allclose(ours, ref, atol=1e-9), q
        U, S = plmc.init_lmc_coefficients(Y, q, QR_form=True)
        Ur, Sr = pj.svd_init(Y, q, QR_form=True)
        assert torch.allclose(U, Ur, atol=1e-9) and torch.allclose(S, Sr, atol=1e-9)


def test_constructor_errors_and_warnings():
    X, Y = _data()
    with pytest.warns(UserWarning, match="dimension of the likelihood"):
        plmc.ProjectedGPModel(X, Y, 6, 2, mean_type=plmc.ZeroMean)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(ValueError, match="non-zero output-wise means"):
            plmc.ProjectedGPModel(X, Y, 6, 2)                      # default ConstantMean is rejected (:927-928)
    with pytest.raises(ValueError, match="Wrong dimensions for Q_plus"):
        plmc.LMCMixingMatrix(torch.eye(5)[:, :3], torch.eye(2))
    with pytest.raises(RuntimeError, match="Likelihood must be Gaussian"):
        plmc.ProjectedLMCmll(torch.nn.Identity(), None)
    with pytest.raises(ValueError, match="prior width"):
        plmc.handle_covar_(plmc.RBFKernel, 3, prior_scales=torch.ones(3))


def test_state_dict_names_are_the_reference_ones():
    X, Y = _data()
    m = _model(X, Y, 2, BDN=True, scalar_B=True, diagonal_B=True, outputscales=True)
    keys = set(m.state_dict())
    assert {"likelihood.noise_covar.raw_noise", "covar_module.base_kernel.raw_lengthscale",
            "covar_module.raw_outputscale", "lmc_coefficients.H", "parametrizations.log_B_tilde.original",
            "train_y", "Y_squared_norm"} <= keys
    m2 = _model(X, Y, 2, BDN=False, bulk=False)
    keys2 = set(m2.state_dict())
    assert {"lmc_coefficients.parametrizations.Q_plus.original", "lmc_coefficients.parametrizations.R.original",
            "parametrizations.B_tilde_inv_chol.original", "M"} <= keys2
    assert m.covar_module.base_kernel.raw_lengthscale.shape == (2, 1, 3)
    assert m.likelihood.noise_covar.raw_noise.shape == (2, 1)
    assert abs(float(m.likelihood.noise[0]) - (0.6931471805599453 + 1.2340980408667956e-4)) < 1e-12


def test_train_mode_requires_training_inputs_and_hot_path_needs_gpu():
    X, Y = _data()
    m = _model(X, Y, 2)
    m.train()
    with pytest.raises(RuntimeError, match="train on the training inputs"):
        m(X[:10])
    out = m(X)
    assert out.mean.shape == (2, 60) and out.lazy_covariance_matrix.shape == (2, 60, 60)
    mll = plmc.ProjectedLMCmll(m.likelihood, m)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mll(out, Y)


def test_c_abi_exports_every_declared_symbol(repo_root):
    from projectedlmc import _hip
    header = open(os.path.join(repo_root, "include", "plmc.h")).read()
    declared = set(re.findall(r"\b(plmc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_hip.exported_symbols())
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.plmc_block() == 128 and lib.plmc_version() == _hip.ABI_VERSION == 4
    lib.plmc_pad.restype, lib.plmc_pad.argtypes = ctypes.c_int64, [ctypes.c_int64]
    assert lib.plmc_pad(1) == 128 and lib.plmc_pad(128) == 128 and lib.plmc_pad(8193) == 8320


def test_lengthscale_priors_match_torch_distributions():
    """NormalPrior / MultivariateNormalPrior (projected_lmc.py:140-149) against torch.distributions; the kernel
    factory starts the lengthscales at the prior mean and registers one prior per sub-kernel."""
    import projectedlmc as plmc
    from projectedlmc.priors import named_priors
    from oracle import priors as opr
    ps = torch.tensor([0.5, 0.7, 0.9], dtype=torch.float64)
    pw = torch.tensor([0.2, 0.3, 0.4], dtype=torch.float64)
    k = plmc.handle_covar_(plmc.RBFKernel, 3, n_funcs=2, prior_scales=ps, prior_width=pw, outputscales=True).double()
    assert torch.allclose(k.base_kernel.lengthscale.detach()[0, 0], ps)
    (entry,) = named_priors(k)
    lp = entry[2].log_prob(entry[3])
    ref = torch.distributions.MultivariateNormal(ps, covariance_matrix=torch.diag(ps * pw)).log_prob(entry[3])
    assert lp.shape == (2, 1) and torch.allclose(lp, ref)
    assert torch.allclose(lp.sum(), opr.lengthscale_log_prior(entry[3], ps, pw))
    k2 = plmc.handle_covar_(plmc.RBFKernel, 2, decomp=[[0], [1]], prior_scales=ps[:2], prior_width=pw[:2]).double()
    pri = named_priors(k2)
    assert len(pri) == 2
    for i, (_, _, pr, val) in enumerate(pri):
        ref = torch.distributions.Normal(ps[i], ps[i] * pw[i]).log_prob(val)
        assert torch.allclose(pr.log_prob(val), ref)


def test_training_input_check_is_remembered_per_tensor_version(monkeypatch):
    """An equal copy of the training inputs is compared once (torch.equal is a host sync on device tensors); the verdict
    is kept for that tensor object at that version, an in-place change brings the comparison -- and the error -- back."""
    X, Y = _data()
    m = _model(X, Y, 2)
    m.train()
    calls = []
    real_equal = torch.equal
    monkeypatch.setattr(torch, "equal", lambda a, b: (calls.append(1), real_equal(a, b))[1])
    Xc = X.clone()
    m(Xc); m(Xc); m(Xc)
    assert len(calls) == 1
    m(X)                                       # the training tensor itself: never compared
    assert len(calls) == 1
    Xc[0, 0] += 1.0                            # version bump -> compared again -> different now
    with pytest.raises(RuntimeError, match="train on the training inputs"):
        m(Xc)
    assert len(calls) == 2
    Xc[0, 0] -= 1.0
    m(Xc)
    assert len(calls) == 3


@pytest.mark.parametrize("shape", [(6, 6), (7, 3)])
def test_reduced_qr_backward_formula_matches_torch(shape):
    """The closed form behind projectedlmc._qr.SmallQR.backward (the device kernel only provides Q and R) against
    autograd through torch.linalg.qr, with both outputs, with R only and with Q only."""
    from projectedlmc import _qr
    g = torch.Generator().manual_seed(4)
    A = torch.randn(*shape, generator=g)
    cQ = torch.randn(*shape, generator=g)
    cR = torch.randn(shape[1], shape[1], generator=g).triu()
    for use_q, use_r in ((True, True), (False, True), (True, False)):
        Ar = A.clone().requires_grad_()
        Q, R = torch.linalg.qr(Ar)
        loss = (Q * cQ).sum() * float(use_q) + (R * cR).sum() * float(use_r)
        loss.backward()
        got = _qr.qr_backward(Q.detach(), R.detach(), cQ if use_q else None, cR if use_r else None)
        assert torch.allclose(got, Ar.grad, rtol=1e-10, atol=1e-12), (use_q, use_r)
    assert _qr.qr_backward(Q.detach(), R.detach(), None, None) is None


def test_deferred_pivot_check_context_restores_the_outer_one():
    from projectedlmc import _engine
    assert _engine.deferred_pivot_checks.current is None
    with _engine.deferred_pivot_checks(1e-6) as outer:
        assert _engine.deferred_pivot_checks.current is outer and outer.jitter == 1e-6
        with _engine.deferred_pivot_checks() as inner:
            assert _engine.deferred_pivot_checks.current is inner
        assert _engine.deferred_pivot_checks.current is outer
    assert _engine.deferred_pivot_checks.current is None
    assert outer.failed() is False and outer.first_bad is None


# ------------------------------------------------------------------ round-2 host fixes
def test_qr_cache_lives_only_inside_a_training_step():
    """The shared QR of the mixing matrix is stored by project_data in training mode and nowhere else: eval-mode calls
    and projection_matrix factor afresh (in-place edits through H.data are seen), and a model that has projected data
    still deep-copies (the step-local caches hold non-leaf tensors and are excluded from the module state)."""
    import copy
    X, Y = _data()
    m = _model(X, Y, 2, init_lmc_coeffs=True, **VARIANTS["PLMC"])
    lmc = m.lmc_coefficients
    m.eval()
    m.project_data(Y)
    m.projection_matrix()
    from projectedlmc.projected import _cache_get
    assert _cache_get(lmc, "qr") is None and _cache_get(m, "QtY") is None
    T0 = m.projection_matrix().detach().clone()
    with torch.no_grad():
        lmc.H.data.mul_(2.0)                              # no version bump
    assert torch.allclose(m.projection_matrix(), T0 / 2.0, atol=1e-12)
    m.train()
    m.project_data(Y)
    assert _cache_get(lmc, "qr") is not None and _cache_get(m, "QtY") is not None
    m2 = copy.deepcopy(m)                                 # used to raise: cached non-leaf tensors in the module dict
    assert _cache_get(m2.lmc_coefficients, "qr") is None and _cache_get(m2, "QtY") is None
    Q1, R1, _ = lmc.QR(reuse=True)
    assert torch.equal(Q1, _cache_get(lmc, "qr")[1][:, :2])
    lmc.drop_qr_cache()
    with torch.no_grad():
        m.project_data(Y)
    assert _cache_get(lmc, "qr") is None                  # no graph, nothing stored


def test_latent_shard_validation():
    X, Y = _data()
    m = _model(X, Y, 2, **VARIANTS["PLMC_fast"])
    m.set_latent_shard((1, 2))
    assert m.latent_ids == [1]
    with pytest.raises(ValueError, match="ranks for 2 latent"):
        m.set_latent_shard((0, 3))                        # a rank with no latent would block the all-reduce
    with pytest.raises(ValueError, match="0 <= rank < world"):
        m.set_latent_shard((2, 2))


def test_workspace_eviction_waits_for_the_gradient_stream(monkeypatch):
    """A workspace dropped from the cache (eviction or free_workspaces) may still be read by the fused K^-1 + gradient
    kernel on the gradient stream: the releasing stream must wait for its `pending` event first."""
    from projectedlmc import _engine

    class FakeStream:
        def __init__(self):
            self.waited = []

        def wait_event(self, e):
            self.waited.append(e)

    class FakeWs:
        def __init__(self, ev):
            self.pending, self.device = ev, torch.device("cpu")

    st = FakeStream()
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: st)
    monkeypatch.setattr(_engine, "_ws_cache", {"a": FakeWs("ev_a"), "b": FakeWs(None), "c": FakeWs("ev_c")})
    _engine.free_workspaces()
    assert st.waited == ["ev_a", "ev_c"] and _engine._ws_cache == {}


def test_stream_handle_carries_its_device():
    from projectedlmc import _hip

    class S:
        cuda_stream = 1234

    h = _hip.stream_handle(S(), torch.device("cuda", 1))
    assert isinstance(h, ctypes.c_void_p) and h.value == 1234 and h.device.index == 1
