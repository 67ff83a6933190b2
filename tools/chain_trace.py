"""Dev aid: where the critical workgroup of the resident chain kernel (csrc/potrf.hip, k_chain) spends a block row.  Needs a variant
build with -DPLMC_CHAIN_TRACE:
    make -C projected-lmc_amd/csrc BUILD=build_trace TARGET=$PWD/tools/variants/libplmc_trace.so EXTRA=-DPLMC_CHAIN_TRACE
    PLMC_LIB=tools/variants/libplmc_trace.so python tools/chain_trace.py [q]
One group (n = 1024), chain alone on the device; stamps are the 100 MHz wall clock: 9 per block row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc import _engine, _hip
q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n, d = 1024, 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
X = (2 * torch.rand(n, d, generator=g) - 1).to(dev)
y = torch.randn(q, n, generator=g).to(dev)
ell = torch.full((q, d), 0.7, device=dev)
noise = torch.full((q,), 0.1, device=dev)
ws = _engine.Workspace(n, q, 1, torch.float32, dev, True)
for rep in range(3):
    _engine.factorize("matern52", X, ell, None, noise, y.reshape(q, 1, n), ws)
    torch.cuda.synchronize()
NB, GMAX = 128, 8
LDG = (GMAX + 1) * NB
m = ws.m
Wg = ws.Vd[0].reshape(-1)[m * NB * NB:]
names = ["waitD", "diag", "postD", "waitP", "loopP", "fuse", "postP", "wbU", "postU"]
rows = []
st = []
for k in range(1 + 9 * 8):
    off = (2 + k // 32) * LDG + GMAX * NB + 2 * (k % 32)
    st.append(int(Wg[off:off + 2].view(torch.int64)[0]))
print("info", ws.info.tolist())
base = st[0]
tot = {nm: 0.0 for nm in names}
k = 1
for r in range(8):
    line = []
    for nm in names:
        if r == 7 and nm in names[3:]:
            break
        dt = (st[k] - st[k - 1]) / 100.0
        tot[nm] += dt
        line.append("%s %.1f" % (nm, dt))
        k += 1
    print("row %d: " % r + "  ".join(line))
print("total %.1f us; per phase: " % ((st[k - 1] - base) / 100.0) + "  ".join("%s %.1f" % (nm, tot[nm]) for nm in names))
