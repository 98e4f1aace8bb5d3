"""configs[3] per-GPU shard: 32768 positions x 256^2 (16 GiB farplane): adjoint identity + timing."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
R1, R2 = 64, 512     # 64 raster rows x 512 columns = one row band of the 512x512 raster
nz, n = syn.object_size_for(R1, R2, 8, 256)
rng = np.random.default_rng(1)
psi_h = syn.random_object(nz, n, rng); scan_h = syn.raster_scan(R1, R2, 8, rng); prb_h = syn.gaussian_probe(256)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.PtychoCuFFT(R1*R2,256,256,1,nz,n)
psi,scan,prb = D(psi_h),D(scan_h),D(prb_h)
gen = torch.Generator(device="cuda").manual_seed(5)
y = torch.view_as_complex(torch.randn((1, R1*R2, 256, 256, 2), generator=gen, device="cuda", dtype=torch.float32))
Ax = slv.fwd(psi, scan, prb)
def dot(a,b):
    s = 0
    for i in range(0, a.shape[1], 4096):
        s = s + torch.sum(a[:, i:i+4096].to(torch.complex128) * b[:, i:i+4096].conj().to(torch.complex128))
    return complex(s)
lhs = dot(Ax, y)
Aty = slv.adj(y, scan, prb); rhs1 = complex(torch.sum(psi.to(torch.complex128)*Aty.conj().to(torch.complex128)))
Bty = slv.adj_probe(y, scan, psi); rhs2 = complex(torch.sum(prb.to(torch.complex128)*Bty.conj().to(torch.complex128)))
print("object", nz, n, "residuals", abs(lhs-rhs1)/abs(lhs), abs(lhs-rhs2)/abs(lhs))
def T(f,n=3):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
del Ax
tf = T(lambda: slv.fwd(psi,scan,prb)); ta = T(lambda: slv.adj(y,scan,prb))
print("fwd ms", tf, "adj ms", ta, "patterns/s pair", R1*R2/((tf+ta)*1e-3))
# CG on the same shard (data = |fwd|^2, 10 iterations from a flat object, position correction on)
del y, Aty, Bty
torch.cuda.empty_cache()
cg = pt.CGPtychoSolver(R1*R2,256,256,1,nz,n); cg.verbose = False
data = torch.empty((1, R1*R2, 256, 256), dtype=torch.float32, device='cuda')
for i in range(0, R1*R2, 4096):
    g = slv.fwd(psi, scan, prb) if i == 0 else g
    data[:, i:i+4096] = torch.abs(g[:, i:i+4096])**2
del g
slv.free()
cg.run(data, torch.ones_like(psi), scan.clone(), prb[None].clone().reshape(1,1,256,256), piter=2); torch.cuda.synchronize()
t=time.perf_counter()
out = cg.run(data, torch.ones_like(psi), scan.clone(), prb[None].clone().reshape(1,1,256,256), piter=10); torch.cuda.synchronize()
dt=(time.perf_counter()-t)/10
print("CG 32768 positions: %.1f ms/iter, %.2f it/s, psi finite %s" % (dt*1e3, 1/dt, bool(torch.isfinite(out['psi']).all())))
