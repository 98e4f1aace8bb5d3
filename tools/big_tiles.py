"""ndet = 1024 / 2048 (no LDS window: direct gather / atomics): operator timings."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
def T(f, n=3):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for ndet, R in ((1024, 32), (2048, 16)):
    nz, n = syn.object_size_for(R, R, 8, ndet)
    rng = np.random.default_rng(1234)
    psi = torch.as_tensor(syn.random_object(nz, n, rng), device='cuda'); scan = torch.as_tensor(syn.raster_scan(R, R, 8, rng), device='cuda')
    prb = torch.as_tensor(syn.gaussian_probe(ndet), device='cuda').contiguous()
    slv = pt.PtychoHIP(R*R, ndet, ndet, 1, nz, n)
    slv.profile(True)
    tf = T(lambda: slv.fwd(psi, scan, prb)); g = slv.fwd(psi, scan, prb)
    ta = T(lambda: slv.adj(g, scan, prb)); tp = T(lambda: slv.adj_probe(g, scan, psi))
    opb = 8.0*R*R*ndet*ndet
    print(ndet, R*R, "positions: fwd %.2f ms adj %.2f ms adj_probe %.2f ms, pair %.1f %% of 8 TB/s" % (tf, ta, tp, 2*opb/((tf+ta)*1e-3)/8e12*100))
    pr = slv.profile_read()
    print("   ", {k: round(ms/c, 2) for k,(ms,c) in pr.items()})
    slv.free(); del g; torch.cuda.empty_cache()
