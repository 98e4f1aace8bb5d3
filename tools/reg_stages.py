"""Stage timings of the fused position registration (config 2)."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
from libtike.hipfft import ptycho as P
from libtike.hipfft import _native as nat
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
dpsi = (psi*0.1+0.01).contiguous()
def T(f, n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("position_shifts total ms", T(lambda: slv._position_shifts(psi, dpsi, 0.01, scan, prb[:,None])))
ones = prb*0+1
print("2 fwd cols", T(lambda: (slv._cg_fwd_cols(0, psi, scan, ones), slv._cg_fwd_cols(1, dpsi, scan, ones))))
ip = torch.empty((4096,256,256), dtype=torch.complex64, device='cuda')
print("cross", T(lambda: nat.check(nat.cg_cross(slv._h, 0, 1, 0.01, P._ptr(ip), P._stream()))))
best = torch.empty(4096, dtype=torch.int64, device='cuda')
print("argmax", T(lambda: nat.check(nat.cg_argmax(slv._h, 1, P._ptr(best), P._stream()))))
idx = 0xffffffff - (best & 0xffffffff)
maxima = torch.stack((idx // 256, idx % 256), dim=1)
print("finish", T(lambda: P._finish_registration(ip, maxima, 100)))
off = torch.rand(4096,2,dtype=torch.float64,device='cuda')*100
print("zoom dft", T(lambda: P._upsampled_dft_batch(ip, 150, 100, off, conj=True)))
cross = P._upsampled_dft_batch(ip, 150, 100, off, conj=True)
print("abs+argmax", T(lambda: P._argmax2d(torch.abs(cross))))
Lc, Rc = P._zoom_factors(256,150,100,1.0,ip.device)
ph = torch.exp(2j*np.pi*torch.rand(4096,256,dtype=torch.float64,device='cuda')).to(torch.complex128)
print("phase mul", T(lambda: torch.mul(ip, ph[:,None,:])))
x = torch.mul(ip, ph[:,None,:])
print("mm1", T(lambda: torch.matmul(x, Rc.T)))
tmp = torch.matmul(x, Rc.T)
print("mul_", T(lambda: tmp.mul_(ph[:,:,None])))
print("bmm core", T(lambda: torch.matmul(Rc, tmp)))
core = torch.matmul(Rc, tmp)
print("expand", T(lambda: torch.matmul(torch.matmul(Lc, core), Lc.T)))
