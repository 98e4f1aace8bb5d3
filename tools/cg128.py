"""CG iteration time at 4096 x 128^2 (the detector size of the reference's own tests), phase-screened probe: native loop, per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
ndet = int(sys.argv[1]) if len(sys.argv) > 1 else 128
p = syn.make_problem(64, 64, 8, ndet, ndet, seed=1234)
D = lambda x: torch.as_tensor(x, device="cuda")
rng = np.random.default_rng(3)
probe = (p["probe"] * np.exp(2j * np.pi * rng.random((ndet, ndet)))).astype(np.complex64)
psi, scan, prb = D(p["psi"]), D(p["scan"]), D(probe)
slv = pt.CGPtychoSolver(4096, ndet, ndet, 1, p["nz"], p["n"]); slv.verbose = False
data = (torch.abs(slv.fwd(psi, scan, prb)) ** 2).contiguous()
for rec in (False, True):
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=10, recover_prb=rec); torch.cuda.synchronize()
    t = time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=50, recover_prb=rec); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 50
    print("ndet %d, 4096 positions, recover_prb=%s: %.3f ms/iter (%.1f it/s) without the in-library profiler" % (ndet, rec, dt * 1e3, 1 / dt))
    slv.profile(True)
    t = time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=30, recover_prb=rec); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 30
    prof = slv.profile_read(); slv.profile(False)
    print("   with an event pair around every launch: %.3f ms/iter; per kernel:" % (dt * 1e3))
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print("   %-28s %7.3f ms/iter %5.1f launches/iter %.3f each" % (k, v[0] / 30, v[1] / 30, v[0] / v[1]))
