"""Multi-mode fused CG (config 2 geometry, 4 modes): per-kernel totals per iteration."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rp = 'prb' in sys.argv
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan = D(p['psi']),D(p['scan'])
prb = D(syn.hermite_modes(256, M))
data = torch.zeros((1,4096,256,256),dtype=torch.float32,device='cuda')
for k in range(M): data += torch.abs(slv.fwd(psi,scan,prb[:,k].contiguous()))**2
slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=2, recover_prb=rp); torch.cuda.synchronize()
N=10
t=time.perf_counter()
slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=N, recover_prb=rp); torch.cuda.synchronize()
wall=(time.perf_counter()-t)/N*1e3
slv.profile(True)
slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=N, recover_prb=rp); torch.cuda.synchronize()
pr = slv.profile_read(); slv.profile(False)
tot=0
for k,(ms,n) in sorted(pr.items(), key=lambda kv:-kv[1][0]):
    print(f"{k:28s} {ms/N:7.3f} ms/iter  {n/N:5.2f} launches/iter  {ms/n:6.3f} ms each"); tot+=ms/N
print(f"{M} modes: kernels {tot:.2f} ms/iter   wall {wall:.2f} ms/iter  -> {1e3/wall:.1f} it/s")
