"""Does the CG rate depend on where the buffers land?  One process, the bench problem, a NEW solver (new work slots)
and new data tensor per repetition, with a dummy allocation of varying size in between to shift the addresses."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
rng = np.random.default_rng(1234)
R, step, ndet = 64, 8, 256
nz, n = syn.object_size_for(R, R, step, ndet)
psi = torch.as_tensor(syn.random_object(nz, n, rng), device='cuda')
scan = torch.as_tensor(syn.raster_scan(R, R, step, np.random.default_rng(1234)), device='cuda')
prb = torch.as_tensor(syn.gaussian_probe(ndet), device='cuda')
keep = []
for rep, pad in enumerate([0, 0, 0, 4096 * 257, 0, 2**20 * 33 + 8192, 0, 2**20 * 7 + 4096 * 3]):
    if pad: keep.append(torch.empty(pad, dtype=torch.uint8, device='cuda'))
    slv = pt.CGPtychoSolver(R*R, ndet, ndet, 1, nz, n); slv.verbose = False
    data = (torch.abs(slv.fwd(psi, scan, prb)) ** 2).contiguous()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=2)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=50)
        torch.cuda.synchronize(); ts.append(50 / (time.perf_counter() - t0))
    print("rep %d pad %9d: %s it/s   data @ %#x" % (rep, pad, " ".join("%.1f" % t for t in ts), data.data_ptr()))
    slv.free(); del data, slv
    torch.cuda.empty_cache()
