"""Host-array wrappers (fwd_ptycho_batch / adj_ptycho_batch) over several angles: seconds per angle."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
NA = 6
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
slv = pt.PtychoHIP(4096,256,256,1,768,768)
psi = np.repeat(p['psi'], NA, axis=0); scan = np.repeat(p['scan'], NA, axis=0); prb = np.repeat(p['probe'], NA, axis=0)
slv.fwd_ptycho_batch(psi[:1], scan[:1], prb[:1])
t=time.perf_counter(); g = slv.fwd_ptycho_batch(psi, scan, prb); tf=(time.perf_counter()-t)/NA
t=time.perf_counter(); f = slv.adj_ptycho_batch(g, scan, prb); ta=(time.perf_counter()-t)/NA
print("fwd_ptycho_batch %.3f s/angle (%.1f GB/s of farplane), adj_ptycho_batch %.3f s/angle (%.1f GB/s)" % (tf, g[0].nbytes/tf/1e9, ta, g[0].nbytes/ta/1e9))
