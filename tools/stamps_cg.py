"""Phase stamps of the column kernels INSIDE the native CG iteration (un-split two-probe forward pass, un-split object adjoint):
diagnostic build (`make -C libtike-cufft_amd/csrc stamps`).  Usage (GPU box): python tools/stamps_cg.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PTYCHO_HIP_LIB", os.path.join(ROOT, "tools", "build", "libptychohip_stamps.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import _native as nat, synthetic as syn

p = syn.make_problem(64, 64, 8, 256, 256, seed=1234, nz=768, n=768)
D = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
rng = np.random.default_rng(3)
probe = (p["probe"] * np.exp(2j * np.pi * rng.random((256, 256)))).astype(np.complex64)
psi, scan, prb = D(p["psi"]), D(p["scan"]), D(probe)
slv = pt.CGPtychoSolver(4096, 256, 256, 1, 768, 768); slv.verbose = False
data = (torch.abs(slv.fwd(psi, scan, prb)) ** 2).contiguous()
dbg = nat.lib.ptycho_debug_stamps
dbg.restype = ctypes.c_int
dbg.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 24)()
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=5); torch.cuda.synchronize()
assert dbg(slv._h, buf) == 0          # allocates + clears
iters = 20
slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=iters); torch.cuda.synchronize()
assert dbg(slv._h, buf) == 0
names = {0: ["loop head (+ probe strip of mode 0)", "gather (LDS taps + weights)", "probe product + step 0 (+ next strip requested)", "exchange store", "barrier A", "window request + exchange load + twiddles + last step", "store issue", "window commit + barrier B", "-", "-", "-", "-"],
         1: ["loop head / probe strip", "wait for tile loads", "transform + probe product", "barrier (prev. combine done)", "window bookkeeping + flush", "barrier (T complete)", "combine", "final flush", "T store", "prefetch issue", "-", "-"]}
for role, title in ((0, "forward column passes of the CG iteration (k_cols_gatherwin<256,FWD,unsplit,NM=2>)"), (1, "k_cols_adjwin<256,unsplit>")):
    v = np.array([buf[12 * role + i] for i in range(12)], dtype=np.float64)
    tot = v.sum()
    print(title, " total wave-cycles per iteration %.3e" % (tot / iters))
    for i in range(12):
        if v[i]:
            print("   %-36s %5.1f %%" % (names[role][i], 100 * v[i] / tot))
