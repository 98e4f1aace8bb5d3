"""Operators and CG at the detector sizes with a mixed-radix plan (48, 80, 96, 112) next to 128 and the Bluestein size 100:
4096 positions, nprb = ndet; one-launch tile kernels (default) and the two-pass kernels; CG iterations per second."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "libtike-cufft_amd")]
import numpy as np, torch
import libtike.hipfft as pt
from libtike.hipfft import synthetic as syn
dev = "cuda"
sizes = [int(a) for a in sys.argv[1:]] or [112, 96, 80, 48, 128, 100]
R, step = 64, 8
for ndet in sizes:
    nz, n = syn.object_size_for(R, R, step, ndet)
    rng = np.random.default_rng(777)
    psi = torch.as_tensor(syn.random_object(nz, n, rng), device=dev)
    scan = torch.as_tensor(syn.raster_scan(R, R, step, rng), device=dev)
    prb = torch.as_tensor(syn.gaussian_probe(ndet), device=dev)
    slv = pt.PtychoCuFFT(R * R, ndet, ndet, 1, nz, n)
    g = torch.empty((1, R * R, ndet, ndet), dtype=torch.complex64, device=dev)
    o = torch.empty_like(psi); op = torch.empty_like(prb)
    def timed(fn, reps=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3): fn()
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    res = []
    for tile in (True, False):
        slv.set_tile(tile)
        res.append((timed(lambda: slv.fwd(psi, scan, prb, out=g)), timed(lambda: slv.adj(g, scan, prb, out=o)), timed(lambda: slv.adj_probe(g, scan, psi))))
    slv.free()
    pair_bytes = 2.0 * (8.0 * R * R * ndet * ndet + 8.0 * nz * n + 8.0 * ndet * ndet + 8.0 * R * R)
    cg = pt.CGPtychoSolver(R * R, ndet, ndet, 1, nz, n); cg.verbose = False
    prs = torch.as_tensor((syn.gaussian_probe(ndet) * np.exp(2j * np.pi * rng.random((ndet, ndet)))).astype(np.complex64), device=dev)
    data = (torch.abs(cg.fwd(psi, scan, prs)) ** 2).contiguous()
    cg.run(data, torch.ones_like(psi), scan.clone(), prs[:, None].clone(), piter=4); torch.cuda.synchronize()
    its = 30
    t0 = time.perf_counter()
    cg.run(data, torch.ones_like(psi), scan.clone(), prs[:, None].clone(), piter=its); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / its
    cg.free()
    (tf, ta, tp), (tf2, ta2, tp2) = res
    print("ndet %4d: tile fwd %.3f adj %.3f adj_probe %.3f ms | two-pass fwd %.3f adj %.3f adj_probe %.3f ms | pair %.3f ms = %.3f of roofline | CG %.2f ms/it = %.0f it/s"
          % (ndet, tf, ta, tp, tf2, ta2, tp2, tf + ta, pair_bytes / ((tf + ta) * 1e-3) / 8e12, dt * 1e3, 1 / dt), flush=True)
