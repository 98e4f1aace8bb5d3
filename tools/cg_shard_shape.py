"""One GPU's share of 4096 positions at 8 GPUs (512 positions), two ways of cutting the 64 x 64 raster:
8 raster rows x 64 columns (a contiguous block of the row-major position list) vs 64 rows x 8 columns."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
D=lambda x: torch.as_tensor(x,device='cuda')
for ny, nx in ((8, 64), (64, 8), (16, 32)):
    p = syn.make_problem(ny, nx, 8, 256, 256, seed=1234)
    slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False
    rng = np.random.default_rng(3)
    probe = (p['probe'] * np.exp(2j*np.pi*rng.random((256,256)))).astype(np.complex64)
    psi,scan,prb = D(p['psi']),D(p['scan']),D(probe)
    data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=300); torch.cuda.synchronize()
    t=time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=200); torch.cuda.synchronize()
    dt = (time.perf_counter()-t)/200*1e3
    slv.profile(True)
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=20); torch.cuda.synchronize()
    prof = slv.profile_read(); slv.profile(False)
    print("%2d rows x %2d columns: %.3f ms/iter; kernels: %s" % (ny, nx, dt, {k: round(v[0]/20, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:6]}))
    slv.free()
