"""Can a column pass at 1 workgroup/CU overlap a memory-bound row pass on another stream?"""
import sys, time, os; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import _native as nat
from libtike.hipfft.ptycho import _ptr
from libtike.hipfft import synthetic as syn
import ctypes
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
D=lambda x: torch.as_tensor(x,device='cuda')
A = pt.CGPtychoSolver(4096,256,256,1,768,768); B = pt.CGPtychoSolver(4096,256,256,1,768,768)
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
data = (torch.abs(A.fwd(psi,scan,prb))**2).contiguous()
A._cg_fwd_cols(0, psi, scan, prb); B._cg_fwd_cols(0, psi, scan, prb)
sums = torch.zeros(2, dtype=torch.float64, device='cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def cols(stream):
    with torch.cuda.stream(stream):
        nat.check(nat.cg_fwd_cols(A._h, 0, _ptr(psi), _ptr(scan), _ptr(prb), ctypes.c_void_p(stream.cuda_stream)))
def rows(stream):
    with torch.cuda.stream(stream):
        nat.check(nat.cg_stats(B._h, 0, _ptr(data), _ptr(sums), ctypes.c_void_p(stream.cuda_stream)))
        nat.check(nat.cg_stats(B._h, 0, _ptr(data), _ptr(sums), ctypes.c_void_p(stream.cuda_stream)))
def T(f,n=10):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("COLSEGS", os.environ.get("PTYCHO_HIP_COLSEGS"), "cols alone", T(lambda: cols(s1)), "2x stats rows alone", T(lambda: rows(s2)),
      "both, two streams", T(lambda: (cols(s1), rows(s2))), "rows then cols same order reversed", T(lambda: (rows(s2), cols(s1))))
