"""configs[4] in miniature: angles streamed through run_batch (host NumPy in/out) vs the same runs on resident data."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
NA, PIT = 12, 20
p = syn.make_problem(64,64,8,256,256,seed=1234,nz=768,n=768)
slv = pt.CGPtychoSolver(4096,256,256,1,768,768); slv.verbose=False
D=lambda x: torch.as_tensor(x,device='cuda')
data1 = (torch.abs(slv.fwd(D(p['psi']),D(p['scan']),D(p['probe'])))**2).cpu().numpy()
data = np.repeat(data1, NA, axis=0); scan = np.repeat(p['scan'], NA, axis=0)
psi0 = np.ones((NA,)+p['psi'].shape[1:], np.complex64); prb = np.repeat(p['probe'][:,None], NA, axis=0)
# resident reference: same work without host traffic
d_gpu, s_gpu, q_gpu = D(data1), D(p['scan']), D(p['probe'][:,None].copy())
slv.run(d_gpu, D(psi0[:1].copy()), s_gpu.clone(), q_gpu.clone(), piter=PIT); torch.cuda.synchronize()
t=time.perf_counter()
for a in range(NA): slv.run(d_gpu, D(psi0[:1].copy()), s_gpu.clone(), q_gpu.clone(), piter=PIT)
torch.cuda.synchronize(); t_res=(time.perf_counter()-t)/NA
t=time.perf_counter()
out = slv.run_batch(data[:NA//2], psi0[:NA//2], scan[:NA//2], prb[:NA//2], piter=PIT)
t_half=time.perf_counter()-t
t=time.perf_counter()
out = slv.run_batch(data, psi0, scan, prb, piter=PIT)
t_full=time.perf_counter()-t
t_batch=(t_full-t_half)/(NA-NA//2)        # marginal cost per angle (setup of the pinned buffers excluded)
print("run_batch %d angles %.2f s, %d angles %.2f s" % (NA//2, t_half, NA, t_full))
print("per angle: resident %.3f s, run_batch (NumPy in/out, %d MiB of data per angle) %.3f s -> overhead %.1f %%" % (t_res, data1.nbytes>>20, t_batch, (t_batch/t_res-1)*100))
