"""dev: soak of the resident chain kernel -- many sweeps at several latent counts and sizes, each compared bit for bit with the first
run of the same problem (a lost hand-over or a stale read would show as a differing factor buffer, an abort as info != 0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine as eng
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for n, q, dtype in ((8192, 8, torch.float32), (8192, 1, torch.float32), (4096, 3, torch.float32), (2304, 5, torch.float64), (8192, 2, torch.float32)):
    g = torch.Generator().manual_seed(n + q)
    d = 8
    X = (2 * torch.rand(n, d, generator=g, dtype=dtype) - 1).to(dev)
    y = torch.randn(q, n, generator=g, dtype=dtype).to(dev)
    ell = torch.linspace(0.4, 1.0, q, dtype=dtype)[:, None].expand(q, d).contiguous().to(dev)
    noise = torch.linspace(0.05, 0.5, q, dtype=dtype).to(dev)
    ws = eng.Workspace(n, q, 1, dtype, dev, True)
    it = torch.int32 if dtype == torch.float32 else torch.int64
    ref = None
    t0 = time.perf_counter()
    for rep in range(reps):
        eng.factorize("matern52", X, ell, None, noise, y.reshape(q, 1, n), ws)
        torch.cuda.synchronize()
        if int(ws.info.abs().max()) != 0:
            print("info != 0 at rep", rep, ws.info.tolist()); bad += 1; break
        cur = (ws.A.view(it).clone(), ws.logdet.clone())
        if ref is None:
            ref = cur
        elif int((cur[0] != ref[0]).sum()) != 0 or not torch.equal(cur[1], ref[1]):
            print("MISMATCH n=%d q=%d rep %d: %d elements" % (n, q, rep, int((cur[0] != ref[0]).sum()))); bad += 1
    print("n=%d q=%d %s: %d sweeps, %.1f ms each, %s" % (n, q, dtype, reps, 1e3 * (time.perf_counter() - t0) / reps, "OK" if bad == 0 else "BAD"), flush=True)
    del ws, ref
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
