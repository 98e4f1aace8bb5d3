"""fwd+adj pair at ndet = 112 (Bluestein path): per-kernel times, windowed object adjoint against the per-pixel atomics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
ndet = int(sys.argv[1]) if len(sys.argv) > 1 else 112
p = syn.make_problem(64, 64, 8, ndet, ndet, seed=3)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
slv = pt.PtychoCuFFT(4096, ndet, ndet, 1, p["nz"], p["n"])
g = slv.fwd(psi, scan, prb)
ref = None
for win in (True, False):
    slv.set_window(win)
    for _ in range(3): o = slv.adj(g, scan, prb)
    slv.profile(True)
    for _ in range(5):
        slv.fwd(psi, scan, prb, out=g); o = slv.adj(g, scan, prb)
    torch.cuda.synchronize()
    prof = slv.profile_read(); slv.profile(False)
    print("window" if win else "atomics", "  ".join("%s %.3f" % (k, ms / c) for k, (ms, c) in prof.items()))
    if ref is None: ref = o.clone()
    else: print("windowed vs atomics rel diff %.2e" % float(torch.abs(o - ref).max() / torch.abs(ref).max()))
