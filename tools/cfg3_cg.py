"""configs[2] CG (4096 x 512^2, 4 Hermite modes): a few iterations, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
R, step, ndet, M = 64, 8, 512, 4
nz = n = 1024
rng = np.random.default_rng(4321)
dev = "cuda"
psi = torch.as_tensor(syn.random_object(nz, n, rng), device=dev)
scan = torch.as_tensor(syn.raster_scan(R, R, step, rng), device=dev)
modes = torch.as_tensor(syn.hermite_modes(ndet, M), device=dev)
slv = pt.CGPtychoSolver(R * R, ndet, ndet, 1, nz, n); slv.verbose = False
data = torch.zeros((1, R * R, ndet, ndet), dtype=torch.float32, device=dev)
for k in range(M):
    g = slv.fwd(psi, scan, modes[:, k].contiguous()); data += torch.abs(g) ** 2
del g; torch.cuda.empty_cache()
slv.run(data, torch.ones_like(psi), scan.clone(), modes.clone(), piter=2); torch.cuda.synchronize()
it = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = time.perf_counter()
slv.run(data, torch.ones_like(psi), scan.clone(), modes.clone(), piter=it); torch.cuda.synchronize()
print("cfg3 CG: %.1f ms/iter" % ((time.perf_counter() - t0) / it * 1e3))
