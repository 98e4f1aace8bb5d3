"""CG iteration time at 4096 x 256^2 with the phase-screened probe (well-conditioned: no deep line searches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64, 64, 8, 256, 256, seed=1234, nz=768, n=768)
D = lambda x: torch.as_tensor(x, device="cuda")
rng = np.random.default_rng(3)
probe = (p["probe"] * np.exp(2j * np.pi * rng.random((256, 256)))).astype(np.complex64)
psi, scan, prb = D(p["psi"]), D(p["scan"]), D(probe)
slv = pt.CGPtychoSolver(4096, 256, 256, 1, 768, 768); slv.verbose = False
data = (torch.abs(slv.fwd(psi, scan, prb)) ** 2).contiguous()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=40); torch.cuda.synchronize()
t = time.perf_counter()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=50); torch.cuda.synchronize()
print("4096 positions, screened probe: %.3f ms/iter" % ((time.perf_counter() - t) / 50 * 1e3))
