import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=3); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=4); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))
