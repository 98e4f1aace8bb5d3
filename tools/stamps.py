"""Where the column kernels spend their cycles: runs the fwd+adj pair of configs[1] on the diagnostic build
(`make -C libtike-cufft_amd/csrc stamps`, in-kernel s_memtime stamps) and prints the share of each phase.
Usage (GPU box): PTYCHO_HIP_LIB=tools/build/libptychohip_stamps.so python tools/stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PTYCHO_HIP_LIB", os.path.join(ROOT, "tools", "build", "libptychohip_stamps.so"))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import _native as nat, synthetic as syn

ND = int(sys.argv[1]) if len(sys.argv) > 1 else 256          # 512: the object adjoint's column pass of configs[2] (un-split kernels)
NO = 768 if ND == 256 else 1024
p = syn.make_problem(64, 64, 8, ND, ND, seed=1234, nz=NO, n=NO)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
psi, scan, prb = dev(p["psi"]), dev(p["scan"]), dev(p["probe"])
slv = pt.PtychoCuFFT(4096, ND, ND, 1, NO, NO)
dbg = nat.lib.ptycho_debug_stamps
dbg.restype = ctypes.c_int
dbg.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 24)()
for _ in range(20 if ND == 256 else 5):
    g = slv.fwd(psi, scan, prb); slv.adj(g, scan, prb)
assert dbg(slv._h, buf) == 0          # allocates + clears
reps = 10
slv.profile(True)
for _ in range(reps):
    g = slv.fwd(psi, scan, prb); slv.adj(g, scan, prb)
torch.cuda.synchronize()
prof = slv.profile_read()
assert dbg(slv._h, buf) == 0
names = {0: ["loop head", "gather (LDS taps + weights)", "probe product + radix-16", "store issue", "barrier A", "window update", "barrier B", "-", "-", "-", "-", "-"],
         1: ["loop head / probe strip", "wait for tile loads", "radix-16 + probe product", "barrier (prev. combine done)", "window bookkeeping + flush", "barrier (T complete)", "combine", "final flush", "T store", "prefetch issue", "-", "-"]}
for role, title in ((0, "k_cols_gatherwin<%d,FWD%s>" % (ND, ",split" if ND == 256 else "")), (1, "k_cols_adjwin<%d%s>" % (ND, ",split" if ND == 256 else ""))):
    v = np.array([buf[12 * role + i] for i in range(12)], dtype=np.float64)
    tot = v.sum()
    print(title, " total wave-cycles per launch %.3e" % (tot / reps))
    for i in range(12):
        if v[i]:
            print("   %-36s %5.1f %%   %8.0f cycles per wave and position" % (names[role][i], 100 * v[i] / tot, v[i] / reps / (4096 * (ND // 16) * (4 if ND == 256 else 8))))
for k, (ms, cnt) in prof.items():
    print("%-24s %.3f ms per launch" % (k, ms / cnt))
