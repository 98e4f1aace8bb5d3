"""the register-resident half-tile forward (tools/probe/half_tile.hip) against ptycho_fwd at
configs[1] (4096 x 256^2, nprb 256): correctness (mode 0) and time of the real kernel and of its ablations.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared [-DGB=8] tools/probe/half_tile.hip -o tools/build/libhalftile.so
    python tools/half_tile_check.py [lib]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np
import torch

import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn

libpath = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "build", "libhalftile.so")
lib = ctypes.CDLL(libpath)
lib.half_tile_fwd.restype = ctypes.c_int
lib.half_tile_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]

p = syn.make_problem(64, 64, 8, 256, 256, seed=1234, nz=768, n=768)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
npos = 4096
slv = pt.PtychoCuFFT(npos, 256, 256, 1, 768, 768)
want = torch.empty((1, npos, 256, 256), dtype=torch.complex64, device="cuda")
got = torch.zeros_like(want)
slv.fwd(psi, scan, prb, out=want)
stream = torch.cuda.current_stream().cuda_stream


def run(mode, grid=256):
    rc = lib.half_tile_fwd(got.data_ptr(), psi.data_ptr(), prb.data_ptr(), scan.data_ptr(), npos, 768, 768, mode, grid, stream)
    assert rc == 0, rc


run(0)
torch.cuda.synchronize()
err = (got - want).abs().max().item() / want.abs().max().item()
print("half-tile forward vs ptycho_fwd: max rel err %.3e" % err, flush=True)
if err > 1e-5:
    d = (got - want).abs()[0]
    bad = (d.amax(dim=(1, 2)) > 1e-5 * want.abs().max()).sum().item()
    print("  positions off:", bad, " first tile rows off:", (d[0].amax(dim=1) > 1e-5 * want.abs().max()).nonzero().flatten()[:16].tolist())

names = {0: "real kernel", 1: "no object/probe loads", 2: "no butterflies", 3: "no loads, no butterflies", 4: "no stores",
         6: "no butterflies, no stores", 7: "LDS transposition + row exchanges only", 10: "no butterflies, no transposition",
         15: "loop skeleton only"}
for rep in range(2):
    for mode in (0, 1, 2, 3, 4, 6, 7, 10, 15):
        for _ in range(2):
            run(mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(mode)
        e1.record()
        torch.cuda.synchronize()
        print("mode %2d  %-42s %.3f ms per 4096 positions" % (mode, names[mode], e0.elapsed_time(e1) / 5), flush=True)
# the two-pass forward for reference
slv.profile(True)
for _ in range(5):
    slv.fwd(psi, scan, prb, out=want)
torch.cuda.synchronize()
print("two-pass forward:", {k: round(ms / c, 3) for k, (ms, c) in slv.profile_read().items()})
