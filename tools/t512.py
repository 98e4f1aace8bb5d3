import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
dev = torch.device('cuda')
R, ndet = 64, 512
rng = np.random.default_rng(4321)
psi = torch.as_tensor(syn.random_object(1024, 1024, rng), device=dev)
scan = torch.as_tensor(syn.raster_scan(R, R, 8, rng), device=dev)
modes = torch.as_tensor(syn.hermite_modes(ndet, 4), device=dev)
slv = pt.CGPtychoSolver(R*R, ndet, ndet, 1, 1024, 1024); slv.verbose = False
prb0 = modes[:, 0].contiguous()
g = slv.fwd(psi, scan, prb0); slv.adj(g, scan, prb0); torch.cuda.synchronize()
slv.profile(True)
t0 = time.perf_counter()
for _ in range(5):
    g = slv.fwd(psi, scan, prb0); slv.adj(g, scan, prb0)
torch.cuda.synchronize(); print("pair ms", (time.perf_counter() - t0) / 5 * 1e3)
print({k: (round(v[0]/v[1], 3), v[1]) for k, v in slv.profile_read().items()}); slv.profile(False)
data = torch.zeros((1, R*R, ndet, ndet), dtype=torch.float32, device=dev)
for k in range(4):
    data += torch.abs(slv.fwd(psi, scan, modes[:, k].contiguous())) ** 2
del g; torch.cuda.empty_cache()
slv.run(data, torch.ones_like(psi), scan.clone(), modes.clone(), piter=2); torch.cuda.synchronize()
slv.profile(True)
t0 = time.perf_counter()
slv.run(data, torch.ones_like(psi), scan.clone(), modes.clone(), piter=4); torch.cuda.synchronize()
print("cg ms/iter", (time.perf_counter() - t0) / 4 * 1e3)
for k, v in sorted(slv.profile_read().items(), key=lambda kv: -kv[1][0]): print("  %-28s %8.3f ms/iter %5.1f launches/iter %.3f each" % (k, v[0]/4, v[1]/4, v[0]/v[1]))
