"""fwd / adj / adj_probe at detector sizes whose tile fits on one CU (ndet 128, 64, 32): per-kernel times and the
fraction of the 8 TB/s roofline on algorithmic bytes (16 ndet^2 B per fwd+adj pattern)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch, time
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
sizes = [int(s) for s in sys.argv[1:]] or [128, 64, 32]
for ndet in sizes:
    R = 64 if ndet >= 64 else 128
    p = syn.make_problem(R, R, 8, ndet, ndet, seed=1234)
    ns = p["nscan"]
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
    slv = pt.PtychoCuFFT(ns, ndet, ndet, 1, p["nz"], p["n"])
    g = torch.empty((1, ns, ndet, ndet), dtype=torch.complex64, device="cuda"); o = torch.empty_like(psi)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o); torch.cuda.synchronize()
    def timed(fn, reps=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    slv.set_tile(False)   # the two-pass kernels of the larger sizes
    two = [timed(lambda: slv.fwd(psi, scan, prb, out=g)), timed(lambda: slv.adj(g, scan, prb, out=o)), timed(lambda: slv.adj_probe(g, scan, psi))]
    slv.set_tile(True)
    tf = timed(lambda: slv.fwd(psi, scan, prb, out=g))
    ta = timed(lambda: slv.adj(g, scan, prb, out=o))
    tp = timed(lambda: slv.adj_probe(g, scan, psi))
    slv.profile(True)
    for _ in range(5): slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
    torch.cuda.synchronize(); prof = slv.profile_read(); slv.profile(False)
    B = 8.0 * ns * ndet * ndet
    print("ndet %4d, %5d positions: fwd %.3f ms (%.3f)  adj %.3f ms (%.3f)  adj_probe %.3f ms | pair %.3f ms = %.3f of 8 TB/s |" %
          (ndet, ns, tf, B / tf / 8e9, ta, B / ta / 8e9, tp, tf + ta, 2 * B / (tf + ta) / 8e9),
          "  ".join("%s %.3f" % (k, ms / c) for k, (ms, c) in prof.items()), "| two-pass fwd %.3f adj %.3f adj_probe %.3f" % tuple(two), flush=True)
    slv.free()
