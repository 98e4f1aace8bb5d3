"""256^2: one shared-gather column pass for two probes (k_cols_gatherwin<256, FWD, false, 2>) against two single passes."""
import sys, ctypes; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import _native as nat, synthetic as syn
from libtike.hipfft.ptycho import _ptr, _stream
p = syn.make_problem(64, 64, 8, 256, 256, seed=1234)
D=lambda x: torch.as_tensor(x,device='cuda')
slv = pt.CGPtychoSolver(p['nscan'],256,256,1,p['nz'],p['n']); slv.verbose=False
psi,scan,prb = D(p['psi']),D(p['scan']),D(p['probe'])
ones = torch.ones_like(prb)
def ev(fn, n=10):
    for _ in range(3): fn()
    a,b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
def two_single():
    nat.check(nat.cg_fwd_cols(slv._h, 0, _ptr(psi), _ptr(scan), _ptr(prb), _stream()))
    nat.check(nat.cg_fwd_cols(slv._h, 1, _ptr(psi), _ptr(scan), _ptr(ones), _stream()))
nat.check(nat.set_option(slv._h, b"compact_modes", 2))
vpp = ctypes.c_void_p * 2
ptrs = vpp(prb.data_ptr(), ones.data_ptr())
def one_shared():
    nat.check(nat.cg_fwd_cols_modes(slv._h, 2, 0, _ptr(psi), _ptr(scan), ptrs, 0, 0, _stream()))
slv._note_scan(scan)
print("two single passes %.3f ms, one shared-gather pass for both %.3f ms" % (ev(two_single), ev(one_shared)))
