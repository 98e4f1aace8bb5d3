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
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=10); torch.cuda.synchronize()
for n in (1, 10, 50, 200):
    t = time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=n); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("piter %4d: %.3f ms total, %.3f ms/iter" % (n, dt * 1e3, dt / n * 1e3), flush=True)
