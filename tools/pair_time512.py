"""fwd+adj pair of configs[2]'s operator (4096 x 512^2, nprb 512): per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64, 64, 8, 512, 512, seed=1234, nz=1024, n=1024)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
slv = pt.PtychoCuFFT(4096, 512, 512, 1, 1024, 1024)
g = torch.empty((1, 4096, 512, 512), dtype=torch.complex64, device="cuda"); o = torch.empty_like(psi)
for _ in range(3): slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
torch.cuda.synchronize()
for rep in range(2):
    slv.profile(True)
    for _ in range(5): slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
    torch.cuda.synchronize()
    prof = slv.profile_read(); slv.profile(False)
    print("  ".join("%s %.3f" % (k, ms / c) for k, (ms, c) in prof.items()), " sum %.2f" % sum(ms / c for ms, c in prof.values()))
