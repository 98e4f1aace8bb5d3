import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
ndet = 112
for (R1, R2) in ((1, 1), (1, 2), (2, 1), (3, 3)):
    p = syn.make_problem(R1, R2, 8, ndet, ndet, seed=3)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
    slv = pt.PtychoCuFFT(p["nscan"], ndet, ndet, 1, p["nz"], p["n"])
    g = slv.fwd(psi, scan, prb)
    slv.set_window(True); a = slv.adj(g, scan, prb).cpu().numpy()[0]
    slv.set_window(False); b = slv.adj(g, scan, prb).cpu().numpy()[0]
    d = np.abs(a - b)
    print(R1, R2, "max rel", d.max() / np.abs(b).max(), "scan", p["scan"][0].tolist())
    bad = d > 1e-4 * np.abs(b).max()
    ys, xs = np.nonzero(bad)
    if len(ys):
        print("  bad rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "count", bad.sum())
        rows = np.unique(ys); cols = np.unique(xs)
        print("  rows:", rows[:40], "cols:", cols[:40])
        y, x = ys[0], xs[0]
        print("  sample", y, x, a[y, x], b[y, x], a[y,x]/b[y,x])
