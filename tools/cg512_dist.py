"""512 positions x 256^2 (one GPU's share of configs[1] under 8-way strong scaling): native CG loop without a process group
against the same loop with a ONE-rank RCCL group (every collective of the N > 1 job is issued; their latency on one rank
is a lower bound of what each costs on eight)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import torch.distributed as dist
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
p = syn.make_problem(16, 32, 8, 256, 256, seed=1234)
D = lambda x: torch.as_tensor(x, device="cuda")
rng = np.random.default_rng(3)
probe = (p["probe"] * np.exp(2j * np.pi * rng.random((256, 256)))).astype(np.complex64)
psi, scan, prb = D(p["psi"]), D(p["scan"]), D(probe)
for name, group in (("no group", None), ("1-rank RCCL group", dist.group.WORLD), ("no group", None), ("1-rank RCCL group", dist.group.WORLD)):
    slv = pt.CGPtychoSolver(p["nscan"], 256, 256, 1, p["nz"], p["n"], group=group); slv.verbose = False
    data = (torch.abs(slv.fwd(psi, scan, prb)) ** 2).contiguous()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=200); torch.cuda.synchronize()
    t = time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=200); torch.cuda.synchronize()
    print("%-18s %.3f ms/iter" % (name, (time.perf_counter() - t) / 200 * 1e3), flush=True)
    slv.free()
dist.destroy_process_group()
