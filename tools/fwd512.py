"""ndet = 512: forward operator with and without the LDS gather window (PTYCHO_HIP_WINDOW=0)."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
R, step, nprb, ndet = 64, 8, 512, 512
nz, n = syn.object_size_for(R, R, step, nprb)
rng = np.random.default_rng(1234)
psi = torch.as_tensor(syn.random_object(nz, n, rng), device='cuda'); scan = torch.as_tensor(syn.raster_scan(R, R, step, rng), device='cuda')
prb = torch.as_tensor(syn.gaussian_probe(nprb), device='cuda').contiguous()
slv = pt.PtychoHIP(R*R, nprb, ndet, 1, nz, n)
def T(f, n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
slv.profile(True)
print("fwd ms", T(lambda: slv.fwd(psi, scan, prb)))
g = slv.fwd(psi, scan, prb)
print("adj ms", T(lambda: slv.adj(g, scan, prb)))
pr = slv.profile_read()
for k,(ms,c) in pr.items(): print(k, round(ms/c,3), "ms each")
