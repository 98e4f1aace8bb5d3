import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(16, 32, 8, 256, 256, seed=1234)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False
rng = np.random.default_rng(3)
probe = (p['probe'] * np.exp(2j*np.pi*rng.random((256,256)))).astype(np.complex64)   # phase screen: well-conditioned line searches
psi,scan,prb = D(p['psi']),D(p['scan']),D(probe)
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=300); torch.cuda.synchronize()   # long warm-up: the clocks of an idle GPU take ~0.3 s to ramp
t=time.perf_counter()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=200); torch.cuda.synchronize()
print("512 positions: %.3f ms/iter" % ((time.perf_counter()-t)/200*1e3), "hints", slv._cg_state[20:22].tolist())
