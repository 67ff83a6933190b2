"""Dev aid: does it pay to run the latents of one step as TWO half batches on two streams (the pipeline fill and drain of one
half's sweep -- chain alone at the start, chain-bound last groups at the end -- beside the bulk of the other half, the K^-1 +
gradient kernel of the first half beside the end of the second half's sweep)?
    python tools/ab_halves.py Q [n]
Times, at n = 8192 (default), d = 8, fp32, the raw pipeline of the exact latent log-prob + gradient (assemble, rhs, sweep,
extract, W^T z, K^-1 + gradient) through the C ABI: (a) one batch of Q latents on one stream, (b) two batches of Q / 2 on two
streams (second one low priority), (c) the same with the second half's enqueue delayed behind the first half's sweep."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, _hip

q = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d = 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, d, generator=g) - 1).to(dev)
y = torch.randn(q, n, generator=g).to(dev)
ell = torch.full((q, d), 0.7, device=dev)
noise = torch.full((q,), 0.7, device=dev)
L = _hip.lib()
dt = torch.float32
KIND = _hip.KIND["matern52"]


_keep = {}


def pipeline(ws, lo, hi, stream, gstream=None):
    """enqueue the whole evaluation of latents lo:hi into workspace ws (q = hi - lo) on `stream`"""
    qq = hi - lo
    if (lo, hi) not in _keep:
        _keep[(lo, hi)] = (ell[lo:hi].contiguous(), noise[lo:hi].contiguous(), y[lo:hi].reshape(qq, 1, n).contiguous())
        torch.cuda.synchronize()
    ell_h, noise_h, y_h = _keep[(lo, hi)]
    with torch.cuda.stream(stream):
        st = _hip.stream_handle(stream, dev)
        _engine.factorize("matern52", X, ell_h, None, noise_h, y_h, ws)
        L.call("plmc_extract_col", dt, _hip.ptr(ws.A), ws.n_pad, ws.lda, ws.strideA, 0, _hip.ptr(ws.z), _hip.ptr(ws.quad), qq, st)
        L.call("plmc_wt_matvec", dt, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.z), _hip.ptr(ws.alpha), qq, st)
        gs = gstream or stream
        if gstream is not None:
            gstream.wait_stream(stream)
        gst = _hip.stream_handle(gs, dev)
        L.call("plmc_kinv_grad_vd", dt, KIND, _hip.ptr(ws.W), ws.n_pad, ws.ldw, ws.strideW, _hip.ptr(ws.alpha), _hip.ptr(X), n, d,
               _hip.ptr(ell_h), None, _hip.ptr(ws.grad), None, 0, 0, None, _hip.ptr(ws.partials), qq, _hip.ptr(noise_h), _hip.ptr(ws.Vd), gst)


def make_ws(qq):
    ws = _engine.Workspace(n, qq, 1, dt, dev, True)
    ws.grad = torch.zeros(qq, d + 2, dtype=torch.float64, device=dev)
    return ws


h = q // 2
lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
sA = torch.cuda.Stream(dev, priority=-1)
sB = torch.cuda.Stream(dev, priority=0)
gA = torch.cuda.Stream(dev, priority=0)
gB = torch.cuda.Stream(dev, priority=0)
ws_all, ws_a, ws_b = make_ws(q), make_ws(h), make_ws(q - h)


def run_single():
    pipeline(ws_all, 0, q, sA)


def run_halves(own_grad_streams):
    pipeline(ws_a, 0, h, sA, gA if own_grad_streams else None)
    pipeline(ws_b, h, q, sB, gB if own_grad_streams else None)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


variants = [("one batch of %d" % q, run_single), ("two halves, two streams", lambda: run_halves(False)),
            ("two halves, K^-1 on own streams", lambda: run_halves(True))]
res = {k: [] for k, _ in variants}
for r in range(7):
    for k, fn in variants:
        fn(); torch.cuda.synchronize()
        t = timed(fn)
        if r:
            res[k].append(t)
for k, _ in variants:
    print("%-40s %7.2f ms (min %7.2f)" % (k, statistics.median(res[k]), min(res[k])))
ga = torch.cat([ws_a.grad, ws_b.grad])
print("gradients equal to the single batch:", bool(torch.equal(ga, ws_all.grad)), float((ga - ws_all.grad).abs().max()))
