import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
for ndet in (112, 128):
    R, step = 64, 8
    nz, n = syn.object_size_for(R, R, step, ndet)
    rng = np.random.default_rng(777)
    psi = torch.as_tensor(syn.random_object(nz, n, rng), device="cuda")
    scan = torch.as_tensor(syn.raster_scan(R, R, step, rng), device="cuda")
    prb = torch.as_tensor(syn.gaussian_probe(ndet), device="cuda")
    slv = pt.PtychoCuFFT(R * R, ndet, ndet, 1, nz, n)
    g = torch.empty((1, R * R, ndet, ndet), dtype=torch.complex64, device="cuda"); o = torch.empty_like(psi)
    for _ in range(5): slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
    torch.cuda.synchronize(); slv.profile(True)
    for _ in range(10): slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
    torch.cuda.synchronize()
    print(ndet, {k: round(ms / c, 4) for k, (ms, c) in slv.profile_read().items()})
