"""Fused single-launch forward (k_fwd_fused256) against the two-pass path and the oracle; timing."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn, _native as nat
from oracle import ptycho_oracle as op
D = lambda x: torch.as_tensor(np.ascontiguousarray(x), device='cuda')
def setf(s, v): nat.check(nat.set_option(s._h, b"fused", v))
# small: vs oracle, incl. padded probe, skipped and overhanging positions, 2 angles
for (nprb, ntheta) in ((256, 1), (128, 2), (200, 1)):
    p = syn.make_problem(3, 4, 13, nprb, 256, ntheta=ntheta, seed=3)
    scan = p["scan"].copy()
    scan[0, 0] = (-1.5, 3.0)                                # skipped
    scan[0, 1] = (p["nz"] - nprb + 0.5, 2.25)               # overhangs the bottom edge
    scan[0, 2] = (0.75, p["n"] - nprb - 0.5)                # touches the right edge
    rng = np.random.default_rng(5)
    prb = (p["probe"] * np.exp(2j * np.pi * rng.random((nprb, nprb)))).astype(np.complex64)
    want = op.fwd(p["psi"], scan, prb, 256, "double")
    with pt.PtychoCuFFT(p["nscan"], nprb, 256, ntheta, p["nz"], p["n"]) as s:
        for mode in (0, 1, 2):
            setf(s, mode)
            g = s.fwd(D(p["psi"]), D(scan), D(prb)).cpu().numpy()
            print("nprb", nprb, "ntheta", ntheta, "fused", mode, "max rel err vs oracle %.2e" % (np.abs(g - want).max() / np.abs(want).max()),
                  "zeros at skipped:", bool((g[0, 0] == 0).all()), flush=True)
# full size timing
p = syn.make_problem(64, 64, 8, 256, 256, seed=1234, nz=768, n=768)
psi, scan, prb = D(p["psi"]), D(p["scan"]), D(p["probe"])
with pt.PtychoCuFFT(4096, 256, 256, 1, 768, 768) as s:
    ref = None
    for mode in (0, 1, 2):
        setf(s, mode)
        g = s.fwd(psi, scan, prb); torch.cuda.synchronize()
        if ref is None: ref = g.clone()
        err = float(torch.abs(g - ref).max() / torch.abs(ref).max())
        t = time.perf_counter()
        for _ in range(20): g = s.fwd(psi, scan, prb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20 * 1e3
        s.profile(True)
        for _ in range(5): g = s.fwd(psi, scan, prb)
        prof = s.profile_read(); s.profile(False)
        print("fused", mode, "fwd %.3f ms" % dt, "diff vs two-pass %.2e" % err, {k: round(v[0] / v[1], 4) for k, v in prof.items()}, flush=True)
