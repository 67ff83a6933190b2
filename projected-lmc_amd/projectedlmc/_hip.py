"""ctypes binding of the C ABI declared in include/plmc.h (libplmc_hip.so, gfx950).

There is deliberately no fallback: if the shared library is missing or a symbol cannot be
resolved, importing the hot path raises.  Pointers are taken from torch tensors
(`tensor.data_ptr()`), the stream from `torch.cuda.current_stream()`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLMC_LIB: dev override (a variant build of the same library, tools/wb_race_probe.py); never a fallback
LIB_PATH = os.environ.get("PLMC_LIB") or os.path.join(_HERE, "libplmc_hip.so")

ABI_VERSION = 4          # include/plmc.h: plmc_version()
KIND = {"rbf": 0, "matern12": 1, "matern32": 2, "matern52": 3, "spline": 4}

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64

# name -> argtypes, for the per-dtype entry points (suffix added below)
_TYPED = {
    "plmc_assemble": [_I, _P, _I, _I, _P, _P, _P, _P, _L, _L, _I, _P],
    "plmc_write_rhs": [_P, _I, _I, _P, _L, _L, _I, _I, _I, _P],
    "plmc_assemble_cross": [_I, _P, _I, _P, _I, _I, _P, _P, _P, _L, _L, _L, _L, _I, _P],
    "plmc_potrf": [_P, _L, _L, _I, _L, _P, _P, _P, _I, _I, _P],
    "plmc_potrf_ex": [_P, _L, _L, _I, _L, _P, _P, _P, _I, _I, _P, _P],
    "plmc_factorize_ex": [_I, _P, _I, _I, _P, _P, _P, _P, _L, _L, _I, _L, _P, _P, _P, _I, _I, _P, _P],
    "plmc_potrs_aug": [_P, _L, _L, _I, _L, _L, _P, _I, _P],
    "plmc_potrs_aug_kept": [_P, _L, _L, _I, _L, _L, _P, _I, _P, _P],
    "plmc_extract_col": [_P, _L, _L, _L, _I, _P, _P, _I, _P],
    "plmc_wt_matvec": [_P, _L, _L, _L, _P, _P, _I, _P],
    "plmc_w_diag": [_P, _L, _L, _L, _P, _I, _P],
    "plmc_gemm_tn": [_I, _I, _I, _I, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _P],
    "plmc_gemm_tn_tri": [_I, _I, _I, _I, _I, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _P],
    "plmc_kinv_grad": [_I, _P, _L, _L, _L, _P, _P, _I, _I, _P, _P, _P, _P, _L, _L, _P, _P, _I, _P],
    "plmc_kinv_grad_ex": [_I, _P, _L, _L, _L, _P, _P, _I, _I, _P, _P, _P, _P, _L, _L, _P, _P, _I, _P, _P],
    "plmc_kinv_grad_vd": [_I, _P, _L, _L, _L, _P, _P, _I, _I, _P, _P, _P, _P, _L, _L, _P, _P, _I, _P, _P, _P],
    "plmc_grad_tiles": [_I, _P, _L, _L, _L, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "plmc_lmc_assemble": [_I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _L, _P],
    "plmc_lmc_cross": [_I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _L, _L, _L, _P],
    "plmc_lmc_kinv_grad": [_I, _P, _L, _L, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "plmc_kernel_vjp": [_I, _P, _I, _P, _I, _I, _P, _P, _P, _L, _L, _P, _P, _P, _I, _P],
    "plmc_qr_small": [_P, _I, _I, _L, _P, _L, _P, _L, _P],
    "plmc_comm_allreduce_sum": [_P, _L, _P],
    "plmc_posterior_moments": [_P, _L, _L, _L, _I, _P, _P, _I, _P],
    "plmc_mix_posterior": [_P, _P, _P, _I, _I, _I, _c.c_double, _P, _P, _P],
}
_PLAIN = {
    "plmc_version": ([], _I),
    "plmc_block": ([], _I),
    "plmc_pad": ([_L], _L),
    "plmc_vd_blocks": ([_L, _L], _L),
    "plmc_vd_blocks_for": ([_L, _L, _I], _L),
    "plmc_vd_blocks_keep": ([_L, _L], _L),
    "plmc_max_dim": ([], _I),
    "plmc_qr_max": ([], _I),
    "plmc_last_error": ([], _c.c_char_p),
    "plmc_grad_scratch_bytes": ([_L, _I], _L),
    "plmc_grad_scratch_bytes_for": ([_L, _I, _I], _L),
    "plmc_grad_partials_bytes": ([_L, _I], _L),
    "plmc_lmc_grad_len": ([_I, _I, _I], _L),
    "plmc_lmc_grad_scratch_bytes": ([_L, _I, _I, _I], _L),
    "plmc_prof_enable": ([_I], _I),
    "plmc_prof_kernels": ([], _I),
    "plmc_prof_name": ([_I], _c.c_char_p),
    "plmc_prof_collect": ([_P, _P, _P, _P], _I),
    "plmc_prof_mfma_rate": ([_I, _P, _L, _P], _I),
    "plmc_dev_reload_knobs": ([], _I),
    "plmc_comm_unique_id": ([_P], _I),
    "plmc_comm_init": ([_P, _I, _I], _I),
    "plmc_comm_world": ([], _I),
    "plmc_comm_rank": ([], _I),
    "plmc_comm_destroy": ([], _I),
}


def prof_enable(on):
    """on: False / True (every kernel class) or an iterable of class names (plmc_prof_name) to bracket."""
    L = lib().cdll
    if isinstance(on, (list, tuple, set)):
        names = [L.plmc_prof_name(i).decode() for i in range(L.plmc_prof_kernels())]
        mask = 0
        for k in on:
            mask |= 1 << names.index(k)
        return L.plmc_prof_enable(mask << 1)
    return L.plmc_prof_enable(1 if on else 0)


def prof_collect():
    """-> {kernel name: dict(ms, launches, flops, bytes)} accumulated since the last collect."""
    L = lib().cdll
    k = L.plmc_prof_kernels()
    ms, fl, by = (_c.c_double * k)(), (_c.c_double * k)(), (_c.c_double * k)()
    la = (_c.c_int64 * k)()
    L.plmc_prof_collect(ms, la, fl, by)
    return {L.plmc_prof_name(i).decode(): dict(ms=ms[i], launches=la[i], flops=fl[i], bytes=by[i])
            for i in range(k) if la[i] > 0}


class knob:
    """Context manager for a dev knob of the library: set the environment variable, make the library re-read its
    knobs, restore on exit (tests / bench.py: `with _hip.knob("PLMC_SERIAL", "1"): ...`)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        os.environ[self.name] = str(self.value)
        lib().cdll.plmc_dev_reload_knobs()
        return self

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.old
        lib().cdll.plmc_dev_reload_knobs()
        return False


def mfma_rate(dtype=torch.float32, device=None):
    """Measured dense MFMA rate (TFLOP/s) of the device: bare instruction stream, no memory traffic.
    dtype: torch.float32 / torch.float64 (v_mfma_*_16x16x4) or "bf16" (v_mfma_f32_16x16x32_bf16, the split engine's)."""
    L = lib().cdll
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    kind = 2 if dtype == "bf16" else int(dtype == torch.float64)
    sink = torch.empty(4 * 1024 * 256, dtype=torch.float32 if kind == 2 else dtype, device=dev)
    out = _c.c_double(0.0)
    with torch.cuda.device(dev):
        torch.cuda.synchronize(dev)
        rc = L.plmc_prof_mfma_rate(kind, ptr(sink), sink.numel() * sink.element_size(), _c.byref(out))
    if rc != 0:
        raise RuntimeError("plmc_prof_mfma_rate failed: %s" % L.plmc_last_error().decode())
    return out.value


def exported_symbols():
    """Every symbol include/plmc.h declares (used by the CPU-side ABI test)."""
    names = list(_PLAIN)
    for base in _TYPED:
        names += [base + "_f32", base + "_f64"]
    return names


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "projectedlmc: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in _PLAIN.items():
            fn = getattr(self.cdll, name)
            fn.argtypes, fn.restype = args, res
        for base, args in _TYPED.items():
            for suf in ("_f32", "_f64"):
                fn = getattr(self.cdll, base + suf)
                fn.argtypes, fn.restype = args, _I
        if self.cdll.plmc_version() != ABI_VERSION:
            raise RuntimeError("projectedlmc: %s is ABI version %d, this package needs %d (the scratch-size contract of "
                               "plmc_potrf_* / plmc_kinv_grad_* differs) -- rebuild it" % (LIB_PATH, self.cdll.plmc_version(), ABI_VERSION))

    def call(self, base, dtype, *args):
        """Typed entry point.  The library picks its helper streams / ordering events by the CURRENT HIP device, so the
        call runs with the device of its stream argument current (stream_ptr / stream_handle record it; a model moved
        with .to('cuda:1') while cuda:0 is current would otherwise queue device-0 helper streams against device-1
        memory)."""
        suf = "_f32" if dtype == torch.float32 else "_f64"
        fn = getattr(self.cdll, base + suf)
        device = next((a.device for a in reversed(args) if isinstance(a, StreamHandle)), None)
        if device is not None and device.index is not None and device.index != torch.cuda.current_device():
            with torch.cuda.device(device):
                rc = fn(*args)
        else:
            rc = fn(*args)
        if rc != 0:
            raise RuntimeError("%s%s failed (%d): %s" % (base, suf, rc, self.cdll.plmc_last_error().decode()))


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr(t):
    return None if t is None else _c.c_void_p(t.data_ptr())


class StreamHandle(_c.c_void_p):
    """hipStream_t for the C ABI that remembers its device (see _Lib.call)."""
    device = None


def stream_handle(stream, device):
    h = StreamHandle(stream.cuda_stream)
    h.device = torch.device(device) if not isinstance(device, torch.device) else device
    if h.device.index is None:
        h.device = torch.device("cuda", torch.cuda.current_device())
    return h


def stream_ptr(device):
    return stream_handle(torch.cuda.current_stream(device), device)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "projectedlmc (MI355X build): the exact-GP hot path runs only on ROCm device tensors "
                "(got a %s tensor); there is no CPU fallback -- move model and data with .cuda()." % t.device.type)
