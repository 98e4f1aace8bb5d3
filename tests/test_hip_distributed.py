"""The N > 1 path on the GPU kernels: two ``gloo`` ranks share the one GPU of the test box
(RCCL refuses two ranks on one device), each owns half of the scan positions and runs the
FUSED CG loop (``ptycho_cg_*`` kernels, position correction and probe recovery on) with
``CGPtychoSolver(group=...)``; both must end with the object / probe / cost trajectory of a
single-process run over all positions.  The production backend (nccl = RCCL, one rank per
GPU) differs only in the transport of the same all-reduces."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem():
    from oracle import ptycho_oracle as op
    from libtike.hipfft import synthetic as syn
    p = syn.make_problem(6, 6, 6, 32, 32, seed=31)
    rng = np.random.default_rng(4)
    probe = (p["probe"][:, None] * np.exp(2j * np.pi * rng.random((32, 32)))).astype(np.complex64)
    data = (np.abs(op.fwd(p["psi"], p["scan"], probe[:, 0], 32)) ** 2).astype(np.float32)
    return p, probe, data


def _run_rank(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import libtike.hipfft as pt
        from libtike.hipfft.distributed import shard_slice
        p, probe, data = _problem()
        sl = shard_slice(p["nscan"], rank, world)
        dev = torch.device("cuda", 0)
        with pt.CGPtychoSolver(sl.stop - sl.start, 32, 32, 1, p["nz"], p["n"], group=dist.group.WORLD) as slv:
            slv.verbose, slv.log_every = False, 1
            assert slv.fused
            res = slv.run(torch.as_tensor(data[:, sl].copy(), device=dev),
                          torch.ones((1, p["nz"], p["n"]), dtype=torch.complex64, device=dev),
                          torch.as_tensor(p["scan"][:, sl].copy(), device=dev),
                          torch.as_tensor(probe.copy(), device=dev), piter=5, recover_prb=True)
            out[rank] = (res["psi"].cpu().numpy(), res["probe"].cpu().numpy(), list(slv.history))
    finally:
        dist.destroy_process_group()


def test_two_rank_fused_cg_matches_single_process():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_run_rank, args=(2, port, out), nprocs=2, join=True)
    import libtike.hipfft as pt
    p, probe, data = _problem()
    dev = torch.device("cuda", 0)
    with pt.CGPtychoSolver(p["nscan"], 32, 32, 1, p["nz"], p["n"]) as slv:
        slv.verbose, slv.log_every = False, 1
        want = slv.run(torch.as_tensor(data.copy(), device=dev),
                       torch.ones((1, p["nz"], p["n"]), dtype=torch.complex64, device=dev),
                       torch.as_tensor(p["scan"].copy(), device=dev),
                       torch.as_tensor(probe.copy(), device=dev), piter=5, recover_prb=True)
        hist = list(slv.history)
    wpsi, wprb = want["psi"].cpu().numpy(), want["probe"].cpu().numpy()
    for r in (0, 1):
        psi, prb, h = out[r]
        assert np.abs(psi - wpsi).max() < 2e-4 * np.abs(wpsi).max()
        assert np.abs(prb - wprb).max() < 2e-4 * np.abs(wprb).max()
        for a, b in zip(h, hist):
            assert a[:3] == b[:3] and abs(a[3] - b[3]) <= 2e-4 * abs(b[3]), (a, b)
    np.testing.assert_array_equal(out[0][0], out[1][0])      # replicas agree bitwise
