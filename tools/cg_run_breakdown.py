"""50-iteration CG run (config 2): wall time per iteration and the in-library per-kernel totals."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
rp = len(sys.argv) > 1
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=2, recover_prb=rp); torch.cuda.synchronize()
t=time.perf_counter()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=50, recover_prb=rp); torch.cuda.synchronize()
wall=(time.perf_counter()-t)/50*1e3
slv.profile(True)
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=50, recover_prb=rp); torch.cuda.synchronize()
pr = slv.profile_read(); slv.profile(False)
tot=0
for k,(ms,n) in sorted(pr.items(), key=lambda kv:-kv[1][0]):
    print(f"{k:28s} {ms/50:7.3f} ms/iter  {n/50:5.2f} launches/iter  {ms/n:6.3f} ms each"); tot+=ms/50
print(f"kernels {tot:.2f} ms/iter   wall {wall:.2f} ms/iter  -> {1e3/wall:.1f} it/s")
