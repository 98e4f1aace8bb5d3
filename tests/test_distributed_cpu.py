"""The N > 1 path on CPU: two ``gloo`` ranks, scan positions sharded, object / probe
gradients and CG scalars all-reduced by ``CGPtychoSolver(group=...)``.  The operators
are supplied by the NumPy oracle (test double, CPU tensors), so what is exercised is the
product's distributed logic: sharding, the all-reduces, and that every rank ends with the
same object as a single-process run over all positions."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ptycho_oracle as op
from libtike.hipfft import synthetic as syn
from libtike.hipfft.distributed import shard_slice
from libtike.hipfft.ptycho import CGPtychoSolver


class CpuSolver(CGPtychoSolver):
    """CG loop of the product on CPU tensors; fwd/adj/fft2 come from the oracle."""

    def __init__(self, nscan, nprb, ndet, ntheta, nz, n, group=None):
        self._sz = dict(ptheta=ntheta, nz=nz, n=n, nscan=nscan, ndet=ndet, nprb=nprb)
        self._h = None
        self._device = torch.device("cpu")
        self.group = group
        self.history = []
        self.verbose = False
        self.log_every = 1
        self.fused = False      # statement-by-statement loop: the operators here are the oracle's

    ptheta = property(lambda s: s._sz["ptheta"])
    nz = property(lambda s: s._sz["nz"])
    n = property(lambda s: s._sz["n"])
    nscan = property(lambda s: s._sz["nscan"])
    ndet = property(lambda s: s._sz["ndet"])
    nprb = property(lambda s: s._sz["nprb"])

    def free(self):
        pass

    def __del__(self):
        pass

    def fwd(self, psi, scan, probe):
        return torch.from_numpy(op.fwd(psi.numpy(), scan.numpy(), probe.numpy(), self.ndet))

    def adj(self, farplane, scan, probe):
        return torch.from_numpy(op.adj(farplane.numpy(), scan.numpy(), probe.numpy(), self.nz, self.n))

    def adj_probe(self, farplane, scan, psi):
        return torch.from_numpy(op.adj_probe(farplane.numpy(), scan.numpy(), psi.numpy(), self.nprb))

    def fft2(self, x, inverse=False, out=None):
        n2 = x.shape[-1] * x.shape[-2]
        return (torch.fft.ifft2(x) * n2) if inverse else torch.fft.fft2(x)


def problem():
    p = syn.make_problem(6, 6, 5, 16, 16, seed=21)
    rng = np.random.default_rng(3)
    probe = (p["probe"][:, None] * np.exp(2j * np.pi * rng.random((16, 16)))).astype(np.complex64)
    data = np.abs(op.fwd(p["psi"], p["scan"], probe[:, 0], 16)) ** 2
    return p, probe, data.astype(np.float32)


def run_rank(rank, world, port, recover, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        p, probe, data = problem()
        sl = shard_slice(p["nscan"], rank, world)
        slv = CpuSolver(sl.stop - sl.start, 16, 16, 1, p["nz"], p["n"], group=dist.group.WORLD)
        res = slv.run(torch.from_numpy(data[:, sl].copy()), torch.ones((1, p["nz"], p["n"]), dtype=torch.complex64),
                      torch.from_numpy(p["scan"][:, sl].copy()), torch.from_numpy(probe.copy()),
                      piter=4, recover_prb=recover)
        out[rank] = (res["psi"].numpy(), res["probe"].numpy(), [h[3] for h in slv.history])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("recover", [False, True])
def test_two_rank_cg_matches_single_process(recover):
    p, probe, data = problem()
    ref = CpuSolver(p["nscan"], 16, 16, 1, p["nz"], p["n"])
    want = ref.run(torch.from_numpy(data.copy()), torch.ones((1, p["nz"], p["n"]), dtype=torch.complex64),
                   torch.from_numpy(p["scan"].copy()), torch.from_numpy(probe.copy()),
                   piter=4, recover_prb=recover)
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + (1 if recover else 0)
    mp.spawn(run_rank, args=(2, port, recover, out), nprocs=2, join=True)
    for r in (0, 1):
        psi, prb, cost = out[r]
        assert np.abs(psi - want["psi"].numpy()).max() < 2e-4
        assert np.abs(prb - want["probe"].numpy()).max() < 2e-4 * np.abs(want["probe"].numpy()).max()
        np.testing.assert_allclose(cost, [h[3] for h in ref.history], rtol=2e-4)
    # replicas are bitwise identical across ranks (all-reduce gives every rank the same sum)
    np.testing.assert_array_equal(out[0][0], out[1][0])


def test_shard_slice_partitions():
    for n, w in ((4096, 8), (10, 3), (5, 8), (1, 1)):
        idx = np.concatenate([np.arange(n)[shard_slice(n, r, w)] for r in range(w)])
        np.testing.assert_array_equal(idx, np.arange(n))
        sizes = [shard_slice(n, r, w).stop - shard_slice(n, r, w).start for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def test_sharded_adjoint_sums_to_full():
    p, probe, data = problem()
    y = np.random.default_rng(0).standard_normal((1, p["nscan"], 16, 16)).astype(np.complex64)
    full = op.adj(y, p["scan"], probe[:, 0], p["nz"], p["n"], "double")
    parts = sum(op.adj(y[:, shard_slice(p["nscan"], r, 3)], p["scan"][:, shard_slice(p["nscan"], r, 3)],
                       probe[:, 0], p["nz"], p["n"], "double") for r in range(3))
    np.testing.assert_allclose(parts, full, atol=1e-10)
