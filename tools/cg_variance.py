"""Why does the bench CG rate vary run to run?  Same problem as bench.py's CG leg, several runs, step sizes logged."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch, math
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
rng = np.random.default_rng(1234)
R, step, ndet = 64, 8, 256
nz, n = syn.object_size_for(R, R, step, ndet)
psi = torch.as_tensor(syn.random_object(nz, n, rng), device='cuda')
scan = torch.as_tensor(syn.raster_scan(R, R, step, np.random.default_rng(1234)), device='cuda')   # bench.py: rank 0's own generator
prb = torch.as_tensor(syn.gaussian_probe(ndet), device='cuda')
slv = pt.CGPtychoSolver(R*R, ndet, ndet, 1, nz, n); slv.verbose = False
data = (torch.abs(slv.fwd(psi, scan, prb)) ** 2).contiguous()
det = not (len(sys.argv) > 1 and sys.argv[1] == "float")
slv.reproducible = det
for rep in range(5):
    slv.log_every = 32
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    slv.log_every = 1; slv.history = []
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=50)
    idx = [int(round(-math.log2(2 * h[1]))) if h[1] > 0 else -1 for h in slv.history]
    print("%s run %d: %.1f it/s; accepted step index per iteration (logged rerun): %s" % ("det" if det else "float", rep, 50 / dt, idx))
