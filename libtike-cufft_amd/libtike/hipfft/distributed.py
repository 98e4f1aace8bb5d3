"""Multi-GPU glue: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL
over xGMI on MI355X; ``gloo`` for CPU tests).

The reference is single-GPU (no collective anywhere in
``/root/reference/src/cuda/ptychofft.cu``).  The path shards naturally over scan
positions: ``fwd`` rows of the farplane are independent per position
(``kernels.cu:14-16,48-57``), ``adj`` / ``adj_probe`` are sums over positions
(``kernels.cu:73-80,92-93``).  Each rank owns a contiguous block of positions (``data``,
``scan`` and every farplane-sized temporary); ``psi`` and ``probe`` are replicated; the
object / probe gradients and the global scalars of the CG loop are all-reduced by
``CGPtychoSolver(group=...)``.
"""
import os

__all__ = ["shard_slice", "init_from_env"]


def shard_slice(nscan, rank, world):
    """Contiguous block of positions of ``rank``: sizes differ by at most one.  A raster
    stored row-major therefore shards into row bands, which keeps the object rows a
    rank touches (and the all-reduce's useful payload) compact."""
    base, rem = divmod(int(nscan), int(world))
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def init_from_env(backend="nccl"):
    """Join the job described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (as set by
    ``python -m torch.distributed.run``).  Returns ``(rank, local_rank, world, group)``;
    ``group`` is ``None`` for a single process."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local)
    if world == 1:
        return rank, local, world, None
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, local, world, dist.group.WORLD
