"""fwd+adj pair of configs[1]: per-kernel times over a few repetitions (in-library HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch, time
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
nprb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = syn.make_problem(64, 64, 8, nprb, 256, seed=1234, nz=768, n=768)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
slv = pt.PtychoCuFFT(4096, nprb, 256, 1, 768, 768)
g = torch.empty((1, 4096, 256, 256), dtype=torch.complex64, device="cuda"); o = torch.empty_like(psi)
t0 = time.time()
while time.time() - t0 < 0.5:
    slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o); torch.cuda.synchronize()
for rep in range(3):
    slv.profile(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
    e1.record(); torch.cuda.synchronize()
    prof = slv.profile_read(); slv.profile(False)
    print("pair %.3f ms |" % (e0.elapsed_time(e1) / 10), "  ".join("%s %.3f" % (k, ms / c) for k, (ms, c) in prof.items()))
