#!/usr/bin/env python3
"""Benchmark of the ptychography hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic scan positions:
``g = fwd(psi, scan, probe)`` followed by ``adj(g, scan, probe)`` on
``--nscan`` positions per GPU (BASELINE.json configs[1]: 4096 x (256 x 256)
complex64, 1 probe mode), plus -- for N > 1 -- the RCCL all-reduce of the object
update, which is the path's one real exchange step.  Scan positions are sharded
over ranks (weak scaling: 4096 positions per GPU on a common object).

Rank 0 prints ONE JSON line.  ``value`` is whole-job fwd+adj patterns/s with all
inputs resident in HBM.  ``roofline`` prices the fwd+adj pair against the HBM
roofline with the algorithmic bytes of BASELINE.md section 3 and reports live
per-kernel durations (HIP events on the launch stream, taken inside the library);
``cpu_baseline`` times the NumPy/SciPy oracle on a bounded sample on this host's
cores.  Secondary: ``cg_iterations_per_s`` (the reference CG loop on the same
problem, 50 iterations) and, for N > 1, ``cg_strong_iterations_per_s`` (the same 4096
positions split over the ranks: strong scaling of the CG loop).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "libtike-cufft_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ndet", type=int, default=256)
    ap.add_argument("--nprb", type=int, default=256)
    ap.add_argument("--raster", type=int, default=64, help="raster is RxR positions per GPU")
    ap.add_argument("--step-px", type=int, default=8)
    ap.add_argument("--cg-iters", type=int, default=50)   # BASELINE.json configs[1]: 50 CG iterations
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=192)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cg", action="store_true")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the configs[2] secondary figures (4096 x 512^2, 4 modes)")
    ap.add_argument("--cfg3-iters", type=int, default=6)
    ap.add_argument("--no-shard", action="store_true", help="skip the configs[3] shard and configs[4] angle-streaming secondary figures")
    return ap.parse_args()


def cfg3_figures(pt, syn, dev, iters):
    """Secondary figures on BASELINE.json configs[2]: 4096 positions x (512 x 512), 4 probe modes
    (Gaussian x Hermite, SURVEY.md 8d cfg 3), object 1024^2: fwd+adj pair of one mode, its roofline
    fraction, multi-mode CG iterations/s and the device memory the solver holds."""
    R, step, ndet, M = 64, 8, 512, 4
    nz = n = 1024
    rng = np.random.default_rng(4321)
    psi = torch.as_tensor(syn.random_object(nz, n, rng), device=dev)
    scan = torch.as_tensor(syn.raster_scan(R, R, step, rng), device=dev)
    modes = torch.as_tensor(syn.hermite_modes(ndet, M), device=dev)
    torch.cuda.synchronize()
    slv = pt.CGPtychoSolver(R * R, ndet, ndet, 1, nz, n)
    slv.verbose = False
    prb0 = modes[:, 0].contiguous()
    for _ in range(3):      # warm-up: both farplane blocks of the alternating pattern below exist afterwards
        g = slv.fwd(psi, scan, prb0)
        slv.adj(g, scan, prb0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g = slv.fwd(psi, scan, prb0)
        slv.adj(g, scan, prb0)
    torch.cuda.synchronize()
    pair_ms = (time.perf_counter() - t0) / 5 * 1e3
    data = torch.zeros((1, R * R, ndet, ndet), dtype=torch.float32, device=dev)
    for k in range(M):
        g = slv.fwd(psi, scan, modes[:, k].contiguous())
        data += torch.abs(g) ** 2
    del g
    slv.release_scratch()           # the adjoint's intermediate of the pair above: the CG loop never uses it
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()      # from here on: what the solver takes next to the caller's data
    slv.run(data, torch.ones_like(psi), scan.clone(), modes.clone(), piter=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), modes.clone(), piter=iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    free1, _ = torch.cuda.mem_get_info()
    pair_bytes = 2.0 * (8.0 * R * R * ndet * ndet + 8.0 * nz * n + 8.0 * ndet * ndet + 8.0 * R * R)
    slv.free()
    return {"cfg3_workload": "4096 positions x (512x512), 4 probe modes, object 1024x1024 (BASELINE.json configs[2])",
            "cfg3_pair_ms": pair_ms, "cfg3_patterns_per_s": R * R / (pair_ms * 1e-3),
            "cfg3_roofline_frac": pair_bytes / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "cfg3_cg_it_s": 1.0 / dt, "cfg3_cg_ms_per_iteration": dt * 1e3, "cfg3_cg_iters_timed": iters,
            "cfg3_device_gib_in_use": (free0 - free1) / 2.0 ** 30,
            "cfg3_note": "CG: 4 modes, no probe recovery, position correction on; memory = device memory the solver takes during "
                         "the run next to the caller's 4 GiB of data (M + 1 = 5 work slots of 8 GiB, summed intensity 4 GiB, registration)"}


def shard_and_stream_figures(pt, syn, dev, iters=6):
    """Secondary figures for the two multi-GPU entries of BASELINE.json, as far as ONE GPU can measure them:
    configs[3]: one rank's share of the 262144-position job = 32768 positions x 256^2 (a 64 x 512 band of the raster, 16 GiB
    farplane): fwd+adj pair and CG iterations/s (the 8-GPU job adds one object all-reduce per step);
    configs[4]: angles streamed through run_batch (NumPy in / out, pinned double buffer + copy stream): seconds per additional
    angle of 4096 x 256^2 with 20 CG iterations each, against the same solves on resident data."""
    out = {}
    R1, R2, step = 64, 512, 8
    nz, n = syn.object_size_for(R1, R2, step, 256)
    rng = np.random.default_rng(31)
    psi = torch.as_tensor(syn.random_object(nz, n, rng), device=dev)
    scan = torch.as_tensor(syn.raster_scan(R1, R2, step, rng), device=dev)
    prb_h = syn.gaussian_probe(256)
    prb = torch.as_tensor((prb_h * np.exp(2j * np.pi * rng.random(prb_h.shape[-2:]))).astype(np.complex64), device=dev)
    npos = R1 * R2
    slv = pt.CGPtychoSolver(npos, 256, 256, 1, nz, n)
    slv.verbose = False
    g = torch.empty((1, npos, 256, 256), dtype=torch.complex64, device=dev)
    o = torch.empty_like(psi)
    for _ in range(2):
        slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
    torch.cuda.synchronize()
    pair_ms = (time.perf_counter() - t0) / 5 * 1e3
    data = torch.empty((1, npos, 256, 256), dtype=torch.float32, device=dev)
    for i in range(0, npos, 4096):
        data[:, i:i + 4096] = torch.abs(g[:, i:i + 4096]) ** 2
    del g
    torch.cuda.empty_cache()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slv.run(data, torch.ones_like(psi), scan.clone(), prb[:, None].clone(), piter=iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    slv.free()
    del data, psi, scan
    torch.cuda.empty_cache()
    pair_bytes = 2.0 * (8.0 * npos * 256 * 256 + 8.0 * nz * n + 8.0 * 256 * 256 + 8.0 * npos)
    out.update({"cfg4_shard_workload": "one GPU's share of configs[3]: 32768 positions x (256x256), object %dx%d (phase-screened probe)" % (nz, n),
                "cfg4_shard_pair_ms": pair_ms, "cfg4_shard_patterns_per_s": npos / (pair_ms * 1e-3),
                "cfg4_shard_roofline_frac": pair_bytes / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "cfg4_shard_cg_it_s": 1.0 / dt, "cfg4_shard_cg_ms_per_iteration": dt * 1e3})
    # configs[4]: angle streaming
    NA, PIT = 6, 20
    p = syn.make_problem(64, 64, 8, 256, 256, seed=1234, nz=768, n=768)
    prbs = (p["probe"] * np.exp(2j * np.pi * np.random.default_rng(5).random((256, 256)))).astype(np.complex64)
    s2 = pt.CGPtychoSolver(4096, 256, 256, 1, 768, 768)
    s2.verbose = False
    D = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    data1 = (torch.abs(s2.fwd(D(p["psi"]), D(p["scan"]), D(prbs))) ** 2).cpu().numpy()
    data = np.repeat(data1, NA, axis=0)
    scan = np.repeat(p["scan"], NA, axis=0)
    psi0 = np.ones((NA,) + p["psi"].shape[1:], np.complex64)
    prb4 = np.repeat(prbs[:, None], NA, axis=0)
    d_gpu, s_gpu, q_gpu = D(data1), D(p["scan"]), D(prbs[:, None].copy())
    s2.run(d_gpu, D(psi0[:1].copy()), s_gpu.clone(), q_gpu.clone(), piter=PIT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s2.run(d_gpu, D(psi0[:1].copy()), s_gpu.clone(), q_gpu.clone(), piter=PIT)
    torch.cuda.synchronize()
    t_res = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    s2.run_batch(data[:NA // 2], psi0[:NA // 2], scan[:NA // 2], prb4[:NA // 2], piter=PIT)
    t_half = time.perf_counter() - t0
    t0 = time.perf_counter()
    s2.run_batch(data, psi0, scan, prb4, piter=PIT)
    t_full = time.perf_counter() - t0
    s2.free()
    out.update({"cfg5_workload": "angles of 4096 positions x (256x256) streamed through run_batch (NumPy in / out, 1 GiB of data per angle), %d CG iterations per angle" % PIT,
                "cfg5_s_per_angle_streamed": (t_full - t_half) / (NA - NA // 2), "cfg5_s_per_angle_resident": t_res,
                "cfg5_note": "marginal cost per additional angle (set-up of the pinned buffers excluded); 180 angles over 8 GPUs = 22.5 angles per GPU, no collective"})
    return out


def generic_figures(pt, syn, dev):
    """fwd+adj pair and CG at detector sizes that are not a power of two (the reference's tests/test_fsc.py:115-120 crops
    to 112): 4096 positions x (112 x 112) on the mixed-radix plan 7 x 4 x 4 (one-launch tile kernels, device-resident
    fused CG loop), and x (100 x 100) on the Bluestein lines (statement-by-statement CG loop) for comparison."""
    out = {}
    for ndet, key, cg_its in ((112, "generic112_", 30), (100, "bluestein100_", 6)):
        R, step = 64, 8
        nz, n = syn.object_size_for(R, R, step, ndet)
        rng = np.random.default_rng(777)
        psi = torch.as_tensor(syn.random_object(nz, n, rng), device=dev)
        scan = torch.as_tensor(syn.raster_scan(R, R, step, rng), device=dev)
        prb = torch.as_tensor(syn.gaussian_probe(ndet), device=dev)
        slv = pt.PtychoCuFFT(R * R, ndet, ndet, 1, nz, n)
        g = torch.empty((1, R * R, ndet, ndet), dtype=torch.complex64, device=dev)
        o = torch.empty_like(psi)
        for _ in range(5):
            slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            slv.adj(slv.fwd(psi, scan, prb, out=g), scan, prb, out=o)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        slv.free()
        cg = pt.CGPtychoSolver(R * R, ndet, ndet, 1, nz, n)
        cg.verbose = False
        prs = torch.as_tensor((syn.gaussian_probe(ndet) * np.exp(2j * np.pi * rng.random((ndet, ndet)))).astype(np.complex64), device=dev)
        data = (torch.abs(cg.fwd(psi, scan, prs)) ** 2).contiguous()
        cg.run(data, torch.ones_like(psi), scan.clone(), prs[:, None].clone(), piter=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cg.run(data, torch.ones_like(psi), scan.clone(), prs[:, None].clone(), piter=cg_its)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / cg_its
        cg.free()
        pair_bytes = 2.0 * (8.0 * R * R * ndet * ndet + 8.0 * nz * n + 8.0 * ndet * ndet + 8.0 * R * R)
        out.update({key + "pair_ms": ms, key + "patterns_per_s": R * R / (ms * 1e-3),
                    key + "roofline_frac": pair_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    key + "cg_it_s": 1.0 / dt})
    out["generic112_workload"] = ("4096 positions x (112x112), nprb 112: mixed-radix Stockham plan 7 x 4 x 4 (sizes 48, 80, 96, 112, 192 have one): "
                                  "one-launch tile forward, device-resident fused CG loop, phase-screened probe, position correction on")
    out["bluestein100_workload"] = ("4096 positions x (100x100), nprb 100: a size without a plan of its own: Bluestein lines, "
                                    "statement-by-statement CG loop (torch elementwise around the HIP operators)")
    return out


def small_tile_figures(pt, syn, dev):
    """fwd / adj at detector sizes whose tile fits one compute unit's LDS (the reference's own tests run ndet = 128,
    tests/test_adjoint.py:18; configs[0] is 64): the forward operator and the probe adjoint are ONE launch each, the
    column<->row intermediate never reaches HBM (k_tile.hpp).  4096 positions, raster step 8 + jitter."""
    out = {}
    for ndet in (128, 64):
        R, step = 64, 8
        nz, n = syn.object_size_for(R, R, step, ndet)
        rng = np.random.default_rng(778)
        psi = torch.as_tensor(syn.random_object(nz, n, rng), device=dev)
        scan = torch.as_tensor(syn.raster_scan(R, R, step, rng), device=dev)
        prb = torch.as_tensor(syn.gaussian_probe(ndet), device=dev)
        slv = pt.PtychoCuFFT(R * R, ndet, ndet, 1, nz, n)
        g = torch.empty((1, R * R, ndet, ndet), dtype=torch.complex64, device=dev)
        o = torch.empty_like(psi)

        def timed(fn, reps=20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                fn()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        tf = timed(lambda: slv.fwd(psi, scan, prb, out=g))
        ta = timed(lambda: slv.adj(g, scan, prb, out=o))
        slv.set_tile(False)
        tf2 = timed(lambda: slv.fwd(psi, scan, prb, out=g))
        ta2 = timed(lambda: slv.adj(g, scan, prb, out=o))
        slv.free()
        op_bytes = 8.0 * R * R * ndet * ndet + 8.0 * nz * n + 8.0 * ndet * ndet + 8.0 * R * R
        k = "ndet%d_" % ndet
        out.update({k + "fwd_ms": tf, k + "adj_ms": ta, k + "fwd_roofline_frac": op_bytes / (tf * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    k + "pair_roofline_frac": 2 * op_bytes / ((tf + ta) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    k + "two_pass_fwd_ms": tf2, k + "two_pass_adj_ms": ta2})
    out["ndet_small_workload"] = "4096 positions x (128x128) and x (64x64), nprb = ndet: one-launch forward (tile in LDS); *_two_pass_* = option tile 0"
    return out


def cpu_baseline(args, prob):
    """The reference ships no CPU path (``array_module = cp`` only), so the CPU figures are the oracle's
    (SURVEY.md 8d), timed on this host: (1) fwd+adj patterns/s with scipy.fft on all cores -- ``value`` --,
    (2) the same with single-threaded numpy.fft, (3) CG iterations/s of the oracle loop on the first 8 x 8
    positions of the workload, extrapolated linearly in the number of positions to the full batch."""
    import warnings
    import scipy.fft
    from oracle import ptycho_oracle as op
    from oracle import cg_oracle as cgo
    cores = len(os.sched_getaffinity(0))
    ns = min(args.cpu_sample, prob["scan"].shape[1])
    scan = prob["scan"][:, :ns]
    done, t_used = 0, 0.0
    with scipy.fft.set_workers(cores):
        while t_used < 10.0:
            t0 = time.perf_counter()
            g = op.fwd(prob["psi"], scan, prob["probe"], args.ndet)
            op.adj(g, scan, prob["probe"], prob["nz"], prob["n"])
            t_used += time.perf_counter() - t0
            done += ns
    out = {"value": done / t_used, "unit": "patterns/s", "cores": cores, "kind": "port",
           "sample": "%d fwd+adj passes over the first %d positions of the workload "
                     "(oracle/ptycho_oracle.py, complex64, scipy.fft workers=%d), %.1f s"
                     % (done // ns, ns, cores, t_used)}
    # (2) single thread, numpy.fft
    n1 = min(32, ns)
    keep = (op.fft2_unnorm, op.ifft2_unnorm)
    op.fft2_unnorm = lambda x: np.fft.fft2(x, axes=(-2, -1)).astype(x.dtype)
    op.ifft2_unnorm = lambda x: np.fft.ifft2(x, axes=(-2, -1), norm="forward").astype(x.dtype)
    try:
        d1, t1 = 0, 0.0
        while t1 < 3.0:
            t0 = time.perf_counter()
            g = op.fwd(prob["psi"], scan[:, :n1], prob["probe"], args.ndet)
            op.adj(g, scan[:, :n1], prob["probe"], prob["nz"], prob["n"])
            t1 += time.perf_counter() - t0
            d1 += n1
    finally:
        op.fft2_unnorm, op.ifft2_unnorm = keep
    out["single_thread"] = {"value": d1 / t1, "unit": "patterns/s", "cores": 1,
                            "sample": "%d fwd+adj passes over %d positions, numpy.fft, one thread, %.1f s" % (d1 // n1, n1, t1)}
    # (3) CG iterations/s of the oracle loop (reference loop restated in NumPy), 64 positions, extrapolated
    R = prob.get("raster", 64)
    idx = (np.arange(8)[:, None] * R + np.arange(8)[None, :]).ravel()          # an 8 x 8 corner of the raster
    sc = np.ascontiguousarray(prob["scan"][:, idx])
    span = int(np.ceil(sc.max())) + args.nprb + 2
    psi = np.ascontiguousarray(prob["psi"][:, :span, :span])
    ora = cgo.OracleSolver(len(idx), args.nprb, args.ndet, 1, span, span)
    with scipy.fft.set_workers(cores), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        data = (np.abs(ora.fwd(psi, sc, prob["probe"])) ** 2).astype(np.float32)
        t0 = time.perf_counter()
        niter = 2
        ora.run(data, np.ones_like(psi), sc.copy(), prob["probe"][:, None].copy(), piter=niter)
        tcg = time.perf_counter() - t0
    nfull = prob["scan"].shape[1]
    out["cg"] = {"value": niter / tcg * len(idx) / nfull, "unit": "CG iterations/s at %d positions (EXTRAPOLATED)" % nfull,
                 "cores": cores, "measured_it_s_on_sample": niter / tcg,
                 "sample": "%d iterations of oracle/cg_oracle.py on %d positions (8 x 8 corner of the raster, object cropped "
                           "to %d^2), scipy.fft workers=%d, %.1f s; extrapolated linearly in the number of positions"
                           % (niter, len(idx), span, cores, tcg)}
    return out


def main():
    args = parse()
    # Libraries (RCCL prints a version banner) must not share stdout with the one JSON line:
    # route fd 1 to stderr until the result is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local = 0          # rehearsal only: several ranks share one GPU (needs BENCH_BACKEND=gloo)
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        # one process per GPU; "nccl" is RCCL on ROCm.  BENCH_FORCE_DIST=1 exercises the
        # collective path with a single rank (used to rehearse the N > 1 code on one GPU).
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29517")
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    ngpu = world

    import libtike.hipfft as pt
    from libtike.hipfft import synthetic as syn

    R, step, nprb, ndet = args.raster, args.step_px, args.nprb, args.ndet
    nscan = R * R
    # common object covering the raster of all ranks (rank r owns raster rows [rR, (r+1)R))
    nz, n = syn.object_size_for(R * ngpu, R, step, nprb)
    rng = np.random.default_rng(1234)
    psi_h = syn.random_object(nz, n, rng)
    prb_h = syn.gaussian_probe(nprb)
    rng_r = np.random.default_rng(1234 + 17 * rank)
    scan_h = syn.raster_scan(R, R, step, rng_r, y0=float(rank * R * step))
    prob = {"psi": psi_h, "probe": prb_h, "scan": scan_h, "nz": nz, "n": n, "raster": R}

    dev = torch.device("cuda", local)
    psi = torch.as_tensor(psi_h, device=dev)
    prb = torch.as_tensor(prb_h, device=dev)
    scan = torch.as_tensor(scan_h, device=dev)

    slv = pt.CGPtychoSolver(nscan, nprb, ndet, 1, nz, n,
                            group=dist.group.WORLD if dist else None)
    slv.verbose = False
    if args.chunk:
        slv.set_chunk(args.chunk)

    # the outputs are allocated once: the timed region measures the operators, not torch's caching allocator
    g_buf = torch.empty((1, nscan, ndet, ndet), dtype=torch.complex64, device=dev)
    upd_buf = torch.empty((1, nz, n), dtype=torch.complex64, device=dev)
    # N > 1: the object update of step k is all-reduced (RCCL, its own stream) while step k + 1 computes -- two update
    # buffers, a step waits for the reduction that last used its buffer.  Every reduction is finished inside the
    # timed region (fence()).
    upd_bufs = [upd_buf, torch.empty_like(upd_buf)] if dist else [upd_buf]
    pending = [None, None]
    step_no = [0]

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def one_step():
        b = step_no[0] & 1 if dist else 0
        step_no[0] += 1
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        g = slv.fwd(psi, scan, prb, out=g_buf)
        upd = slv.adj(g, scan, prb, out=upd_bufs[b])   # zero-fills its output, as the reference's adj does (ptycho.py:102)
        if dist:
            pending[b] = dist.all_reduce(torch.view_as_real(upd), async_op=True)
        return upd

    # ~0.3 s of untimed work first: an idle MI355X ramps its clocks over that long (tools/cg512.py), and the
    # official warm-up of a few steps is only a few milliseconds.  Local work only: the loop is bounded by each rank's
    # own clock, so a collective inside it would be issued a different number of times per rank (deadlock).
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.4:
        slv.adj(slv.fwd(psi, scan, prb, out=g_buf), scan, prb, out=upd_buf)
        torch.cuda.synchronize()

    def fence():
        drain()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = ngpu * nscan * args.steps / elapsed

    # The timed steps reuse the position order of the first call (the scan tensor is unchanged: option trust_order).
    # A caller whose positions change between calls pays the sort in both operators: same step, order forgotten
    # before each operator call (local work only: no collective in this loop).
    def sorted_step():
        slv._scan_key = None
        g = slv.fwd(psi, scan, prb, out=g_buf)
        slv._scan_key = None
        slv.adj(g, scan, prb, out=upd_buf)
    for _ in range(args.warmup):
        sorted_step()
    torch.cuda.synchronize()
    t0s = time.perf_counter()
    for _ in range(args.steps):
        sorted_step()
    torch.cuda.synchronize()
    ms_per_step_sorting = 1e3 * (time.perf_counter() - t0s) / args.steps

    # ---- live per-kernel timing (HIP events inside the library, launch stream) --
    slv.profile(True)
    nprof = 3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(nprof):
        g = slv.fwd(psi, scan, prb, out=g_buf)
        slv.adj(g, scan, prb, out=upd_buf)
    ev1.record()
    torch.cuda.synchronize()
    prof = slv.profile_read()
    slv.profile(False)
    pair_ms_events = ev0.elapsed_time(ev1) / nprof
    kern = {k: {"avg_ms_per_launch": ms / cnt, "launches_per_step": cnt / nprof,
                "ms_per_step": ms / nprof} for k, (ms, cnt) in prof.items()}
    kernel_ms = sum(v["ms_per_step"] for v in kern.values())
    dominant = max(kern, key=lambda k: kern[k]["ms_per_step"])
    op_bytes = 8.0 * nscan * ndet * ndet + 8.0 * nz * n + 8.0 * nprb * nprb + 8.0 * nscan
    pair_bytes = 2.0 * op_bytes
    achieved = pair_bytes / (1e-3 * ms_per_step) / 1e9
    # the dominant kernel against its own operator's compulsory bytes per launch
    dom = kern[dominant]
    dom_bytes = op_bytes / dom["launches_per_step"]
    # HBM traffic per step from the committed rocprofv3 PMC passes of this same command
    # (profiles/traffic.json; a live PMC read is not possible from inside the process)
    traffic, traffic_src, dom_own = None, None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        names = tj["roles"]     # bench kernel role -> rocprof kernel name of the profiled build
        if ndet == 256 and nprb == 256 and nscan == 4096 and all(names.get(k, "") in tj["kernels"] for k in kern):
            per = {k: tj["kernels"][names[k]].get("fetch_bytes_per_launch", 0.0)
                      + tj["kernels"][names[k]].get("write_bytes_per_launch", 0.0) for k in kern}
            traffic = sum(per.values())
            dom_own = per[dominant]
            traffic_src = tj["source"]
    except Exception:
        pass
    roofline = {
        "bound": "hbm",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_unit": "HBM bytes per step (fwd+adj pair), PMC", "traffic_source": traffic_src,
        "what": "fwd+adj pair: %.4e algorithmic B per %d-position batch (16*ndet^2 B/pattern "
                "+ object, probe, scan) / measured ms_per_step" % (pair_bytes, nscan),
        # the kernel with the longest duration per step.  Two different questions, two keys: (1) what share of the operator's
        # ALGORITHMIC bytes its duration alone would allow ("operator_bytes_over_this_kernel": a ceiling on the operator's own
        # fraction, NOT a kernel efficiency -- the operator has a second kernel); (2) how fast the kernel moves the bytes it
        # really touches ("own_traffic_*", from the PMC passes of profiles/traffic.json: for the row pass the intermediate in
        # and the farplane out)
        "dominant_kernel": {
            "name": dominant, "avg_ms_per_launch": dom["avg_ms_per_launch"],
            "launches_per_step": dom["launches_per_step"],
            "operator_algorithmic_bytes_per_launch": dom_bytes,
            "operator_bytes_over_this_kernel_GBs": dom_bytes / (1e-3 * dom["avg_ms_per_launch"]) / 1e9,
            "operator_bytes_over_this_kernel_frac": dom_bytes / (1e-3 * dom["avg_ms_per_launch"]) / 1e9 / HBM_PEAK_GBS,
            "own_traffic_bytes_per_launch": dom_own,
            "own_traffic_GBs": (dom_own / (1e-3 * dom["avg_ms_per_launch"]) / 1e9) if dom_own else None,
            "own_traffic_frac_of_peak": (dom_own / (1e-3 * dom["avg_ms_per_launch"]) / 1e9 / HBM_PEAK_GBS) if dom_own else None,
        },
        "kernels": kern,
        "kernel_ms_per_step": kernel_ms, "pair_ms_hip_events": pair_ms_events,
    }

    # ---- CG iterations / s (secondary metric of BASELINE.json) -------------------
    cg = None
    if not args.no_cg and args.cg_iters > 0:
        data = (torch.abs(slv.fwd(psi, scan, prb, out=g_buf)) ** 2).contiguous()
        g_buf = None
        psi0 = torch.ones_like(psi)
        slv.run(data, psi0, scan.clone(), prb[:, None].clone(), piter=2)     # warm-up
        fence()
        t0 = time.perf_counter()
        slv.run(data, psi0, scan.clone(), prb[:, None].clone(), piter=args.cg_iters)
        fence()
        dt = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        strong = None
        if dist and ngpu > 1 and R % ngpu == 0:
            try:
                # strong scaling of the CG loop (BASELINE.json target: >= 6x at 8 GPUs): the SAME
                # 4096-position problem of configs[1], raster columns [r R/N, (r+1) R/N) on rank r (column bands keep
                # the runs of the windowed column kernels at full raster height: tools/cg_shard_shape.py, 1.33 vs 1.38 ms)
                del data
                Rr = R // ngpu
                nz1, n1 = syn.object_size_for(R, R, step, nprb)
                psi1 = torch.as_tensor(syn.random_object(nz1, n1, np.random.default_rng(1234)), device=dev)
                scan1 = torch.as_tensor(syn.raster_scan(R, Rr, step, np.random.default_rng(99 + rank),
                                                         x0=float(rank * Rr * step)), device=dev)
                s2 = pt.CGPtychoSolver(Rr * R, nprb, ndet, 1, nz1, n1, group=dist.group.WORLD)
                s2.verbose = False
                data1 = (torch.abs(s2.fwd(psi1, scan1, prb)) ** 2).contiguous()
                s2.run(data1, torch.ones_like(psi1), scan1.clone(), prb[:, None].clone(), piter=2)
                fence()
                t0 = time.perf_counter()
                s2.run(data1, torch.ones_like(psi1), scan1.clone(), prb[:, None].clone(), piter=args.cg_iters)
                fence()
                t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                strong = {"cg_strong_iterations_per_s": args.cg_iters / float(t.item()),
                          "cg_strong_config": "%d positions in total (%d per GPU), same loop" % (R * R, Rr * R)}
                s2.free()
                data = data1
            except Exception as e:      # never let the secondary figure break the primary line
                print("strong-scaling CG run failed: %r" % (e,), file=sys.stderr)
                strong = None
                data = None
        screened = None
        if not dist:
            # the same loop on the oracle-validated family of problems: the probe carries a random phase screen, so the
            # model covers the whole detector from the flat start and the trajectory is well conditioned (the smooth
            # probe's is carried by rounding noise: tests/test_oracle_divergence.py); tests/test_hip_configs.py checks
            # this geometry against the oracle at 8 x 8 positions
            rs = np.random.default_rng(4242)
            prb_s = torch.as_tensor((prb_h * np.exp(2j * np.pi * rs.random(prb_h.shape[-2:]))).astype(np.complex64), device=dev)
            if data is None or g_buf is None:
                g_buf = torch.empty((1, nscan, ndet, ndet), dtype=torch.complex64, device=dev)
            data_s = (torch.abs(slv.fwd(psi, scan, prb_s, out=g_buf)) ** 2).contiguous()
            g_buf = None
            slv.run(data_s, psi0, scan.clone(), prb_s[:, None].clone(), piter=2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            slv.run(data_s, psi0, scan.clone(), prb_s[:, None].clone(), piter=args.cg_iters)
            torch.cuda.synchronize()
            dts = time.perf_counter() - t0
            screened = {"cg_screened_iterations_per_s": args.cg_iters / dts,
                        "cg_screened_config": "same geometry and loop, probe x random phase screen (the oracle-checked, well-conditioned variant)"}
            del data_s
        cg = {"cg_iterations_per_s": args.cg_iters / dt, "cg_iters_timed": args.cg_iters,
              "cg_ms_per_iteration": dt / args.cg_iters * 1e3,
              "cg_config": "gaussian, 1 mode, no probe recovery, position correction on (reference loop, sequenced by "
                           "the native stage calls), initial object = 1, timed from iteration 0"
                           + (" (weak: %d positions per GPU)" % nscan if ngpu > 1 else "")}
        if strong:
            cg.update(strong)
        if screened:
            cg.update(screened)
        del data

    out = {
        "metric": "fwd+adj patterns/s",
        "value": value, "unit": "patterns/s",
        "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_sorting_every_call": ms_per_step_sorting,
        "timed_region_note": "value / ms_per_step: the position order is computed once (scan unchanged between calls); "
                             "ms_per_step_sorting_every_call: the same pair with the order recomputed in fwd and in adj"
                             + (" (no all-reduce in this second loop)" if ngpu > 1 else ""),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%d scan positions/GPU x (%dx%d) detector, complex64, nprb=%d, "
                               "1 probe mode, object %dx%d, raster step %d px + jitter "
                               "(BASELINE.json configs[1])" % (nscan, ndet, ndet, nprb, nz, n, step),
                   "nscan_per_gpu": nscan, "ndet": ndet, "nprb": nprb, "object": [nz, n],
                   "sharding": "scan positions over ranks, object all-reduce (RCCL)" if ngpu > 1 else "single GPU",
                   "chunk": int(args.chunk)},
        "roofline": roofline,
    }
    if cg:
        out.update(cg)
    if rank == 0 and ngpu == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args, prob)
    elif rank == 0:
        out["cpu_baseline"] = None
    slv.free()
    if ngpu == 1 and not dist and not args.no_cg:
        del psi, scan, prb
        torch.cuda.empty_cache()
        if not args.no_cfg3:
            try:
                out.update(cfg3_figures(pt, syn, dev, args.cfg3_iters))
            except Exception as e:          # never let a secondary figure break the primary line
                print("configs[2] figures failed: %r" % (e,), file=sys.stderr)
        try:
            out.update(generic_figures(pt, syn, dev))
            out.update(small_tile_figures(pt, syn, dev))
        except Exception as e:
            print("generic-size figures failed: %r" % (e,), file=sys.stderr)
        if not args.no_shard:
            try:
                torch.cuda.empty_cache()
                out.update(shard_and_stream_figures(pt, syn, dev))
            except Exception as e:
                print("configs[3] shard / configs[4] streaming figures failed: %r" % (e,), file=sys.stderr)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
