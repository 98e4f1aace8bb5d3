"""Timings for BASELINE.json configs other than the bench.py headline (configs[2]: 4096 x 512^2, 4 modes;
a padded-probe variant of configs[1])."""
import sys, time, json; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
def T(f, n=5):
    f(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
def run(name, R, step, nprb, ndet, nmodes, cg_iters):
    nz, n = syn.object_size_for(R, R, step, nprb)
    rng = np.random.default_rng(1234)
    psi_h = syn.random_object(nz, n, rng); scan_h = syn.raster_scan(R, R, step, rng)
    probes = syn.hermite_modes(nprb, nmodes) if nmodes > 1 else syn.gaussian_probe(nprb)[:, None]
    D = lambda x: torch.as_tensor(x, device='cuda')
    slv = pt.CGPtychoSolver(R*R, nprb, ndet, 1, nz, n); slv.verbose = False
    psi, scan, prb = D(psi_h), D(scan_h), D(probes)
    out = {"config": name, "nscan": R*R, "ndet": ndet, "nprb": nprb, "nmodes": nmodes, "object": [nz, n]}
    out["fwd_ms"] = T(lambda: slv.fwd(psi, scan, prb[:, 0].contiguous()))
    g = slv.fwd(psi, scan, prb[:, 0].contiguous())
    out["adj_ms"] = T(lambda: slv.adj(g, scan, prb[:, 0].contiguous()))
    out["adj_probe_ms"] = T(lambda: slv.adj_probe(g, scan, psi))
    opb = 8.0*R*R*ndet*ndet + 8.0*nz*n + 8.0*nprb*nprb + 8.0*R*R
    out["pair_frac_of_8TBs"] = 2*opb/((out["fwd_ms"]+out["adj_ms"])*1e-3)/8e12
    del g
    data = torch.zeros((1, R*R, ndet, ndet), dtype=torch.float32, device='cuda')
    for k in range(nmodes): data += torch.abs(slv.fwd(psi, scan, prb[:, k].contiguous()))**2
    if cg_iters:
        slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=2)
        torch.cuda.synchronize(); t=time.perf_counter()
        slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=cg_iters)
        torch.cuda.synchronize(); out["cg_iter_per_s"] = cg_iters/(time.perf_counter()-t)
        slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=2, recover_prb=True)
        torch.cuda.synchronize(); t=time.perf_counter()
        slv.run(data, torch.ones_like(psi), scan.clone(), prb.clone(), piter=cg_iters, recover_prb=True)
        torch.cuda.synchronize(); out["cg_iter_per_s_recover_prb"] = cg_iters/(time.perf_counter()-t)
    print(json.dumps(out), flush=True)
    slv.free(); del data; torch.cuda.empty_cache()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "pad"): run("configs[1] with nprb=128<ndet=256", 64, 8, 128, 256, 1, 20)
if which in ("all", "c2"): run("configs[1]", 64, 8, 256, 256, 1, 6)
if which in ("all", "c3"): run("configs[2]: 512^2, 4 modes", 64, 8, 512, 512, 4, 10)
