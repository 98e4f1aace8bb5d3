import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
from libtike.hipfft.ptycho import register_translation_batch
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
g = slv.fwd(psi,scan,prb)[0]; g2 = slv.fwd(psi*0.9+0.05,scan,prb)[0]
for _ in range(2): register_translation_batch(g, g2, 100, 'fourier', op=slv)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    register_translation_batch(g, g2, 100, 'fourier', op=slv); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
