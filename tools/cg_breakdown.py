"""Time the stages of one CG iteration at config 2 (torch-glue solver)."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
from libtike.hipfft.ptycho import register_translation_batch
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(slv.fwd(psi,scan,prb))**2).contiguous()
def T(f, n=3):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("fwd ms", T(lambda: slv.fwd(psi,scan,prb)))
g = slv.fwd(psi,scan,prb)
print("adj ms", T(lambda: slv.adj(g,scan,prb)))
print("abs**2 accumulate ms", T(lambda: torch.abs(g)**2))
I = torch.abs(g)**2
print("a,b sums ms", T(lambda: (torch.sum(torch.sqrt(I*data)), torch.sum(I))))
print("projection ms", T(lambda: g - torch.sqrt(data)*g/(torch.sqrt(I)+1e-32)))
g2 = slv.fwd(psi*0.9,scan,prb)
print("p1p2p3 ms", T(lambda: (torch.abs(g)**2, torch.abs(g2)**2, 2*(g.real*g2.real+g.imag*g2.imag))))
p1,p2,p3 = torch.abs(g)**2, torch.abs(g2)**2, 2*(g.real*g2.real+g.imag*g2.imag)
minf = lambda x: torch.sum((torch.sqrt(torch.abs(x))-torch.sqrt(data))**2)
print("one line-search trial ms", T(lambda: float(minf(p1+0.25*p2+0.5*p3))))
print("registration ms", T(lambda: register_translation_batch(g[0], g2[0], 100, 'fourier', op=slv), 2))
print("fft2 ms", T(lambda: slv.fft2(g[0], inverse=True)))
t=time.perf_counter(); slv.run(data, torch.ones_like(psi), scan.clone(), prb[:,None].clone(), piter=4); torch.cuda.synchronize()
print("4 iters ms/iter", (time.perf_counter()-t)/4*1e3)
