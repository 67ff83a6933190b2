"""Race probe for the multi-stream sweep: factor the same matrices repeatedly under the look-ahead schedule and
compare the whole factor buffer (U, augmented column, W) bit for bit between runs and with the one-stream schedule
(PLMC_SERIAL=1).  Mismatching 128 x 128 tiles are listed by (latent, block row, block column, region).

  PLMC_LIB=/path/to/variant.so python tools/wb_race_probe.py [--n 8192 --q 8 --dtype f32 --reps 6]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]

import torch  # noqa: E402

from projectedlmc import _engine, _hip  # noqa: E402


def factor(ws, X, ell, noise, y, kind="matern52"):
    _engine.factorize(kind, X, ell, None, noise, y.reshape(ws.q, 1, ws.n), ws)
    torch.cuda.synchronize()
    return ws.A.clone(), ws.logdet.clone(), ws.info.clone()


def tiles_differ(a, b, ws):
    """-> list of (lat, block row, block col, region) where the buffers differ (NaN-safe bitwise compare)."""
    it = torch.int32 if a.dtype == torch.float32 else torch.int64
    ne = (a.view(it) != b.view(it))
    q, n_pad, lda = a.shape
    nb = ws.NB
    nbc = lda // nb
    t = ne.view(q, n_pad // nb, nb, nbc, nb).any(dim=4).any(dim=2)       # (q, m, nbc)
    out = []
    for lat, r, c in t.nonzero().tolist():
        col = c * nb
        if col < ws.n_pad:
            reg = "U" if c >= r else "U-lower(unused)"
        elif col < ws.wcol0:
            reg = "aug"
        elif col < ws.wcol0 + ws.n_pad:
            c = (col - ws.wcol0) // nb
            reg = "W" if c <= r else "W-upper(unused)"
        else:
            reg = "pad"
        out.append((lat, r, c, reg))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--q", type=int, default=8)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    X = (2 * torch.rand(a.n, a.d, generator=g, dtype=dt) - 1).to(dev)
    y = torch.randn(a.q, a.n, generator=g, dtype=dt).to(dev)
    ell = torch.linspace(0.4, 1.0, a.q, dtype=dt)[:, None].expand(a.q, a.d).contiguous().to(dev)
    noise = torch.linspace(0.05, 0.5, a.q, dtype=dt).to(dev)
    ws = _engine.Workspace(a.n, a.q, 1, dt, dev, True)
    with _hip.knob("PLMC_SERIAL", "1"):
        ref, ld_ref, info_ref = factor(ws, X, ell, noise, y)
        ref2, _, _ = factor(ws, X, ell, noise, y)
    res = {"lib": os.environ.get("PLMC_LIB", _hip.LIB_PATH), "tag": a.tag, "n": a.n, "q": a.q, "dtype": a.dtype,
           "serial_repeatable": bool(torch.equal(ref.view(torch.uint8), ref2.view(torch.uint8))),
           "serial_tiles_differ": len(tiles_differ(ref2, ref, ws)),
           "info_serial": info_ref.tolist(), "runs": []}
    del ref2
    for rep in range(a.reps):
        A, ld, info = factor(ws, X, ell, noise, y)
        bad = tiles_differ(A, ref, ws)
        useful = [b for b in bad if "unused" not in b[3] and b[3] != "pad"]
        detail = None
        if useful:
            lat, r, c, reg = min(useful, key=lambda b: (b[1], b[0], b[2]))
            col0 = c * ws.NB if reg in ("U",) else (ws.n_pad if reg == "aug" else ws.wcol0 + c * ws.NB)
            got = A[lat, r * ws.NB:(r + 1) * ws.NB, col0:col0 + ws.NB]
            exp = ref[lat, r * ws.NB:(r + 1) * ws.NB, col0:col0 + ws.NB]
            ne = got != exp
            rows = ne.any(1).nonzero().flatten().tolist()
            cols = ne.any(0).nonzero().flatten().tolist()
            idx = ne.nonzero()[:6].tolist()
            detail = {"tile": [lat, r, c, reg], "rows": rows, "ncols": len(cols), "cols_head": cols[:8], "cols_tail": cols[-4:],
                      "samples": [(i, j, float(got[i, j]), float(exp[i, j])) for i, j in idx],
                      "n_elems": int(ne.sum())}
        res["runs"].append({"rep": rep, "detail": detail, "info": info.tolist(), "tiles_differ": len(bad), "live_tiles_differ": len(useful),
                            "first": useful[:24], "logdet_equal": bool(torch.equal(ld, ld_ref))})
        del A
    print(json.dumps(res))


if __name__ == "__main__":
    main()
