"""``bench.py`` with more than one rank (the driver's N = 2, 4, 8 runs): two ``gloo`` ranks share the test box's one
GPU (``BENCH_SINGLE_DEVICE=1``; RCCL refuses two ranks on one device) and run the whole script on a small raster --
weak-scaling step with its object all-reduce, CG leg, strong-scaling CG leg.  Every collective must be issued the
same number of times on both ranks: a wall-clock-bounded loop around one (as the clock-ramp loop of round 2 once
was) hangs here, and the time limit turns that into a failure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_with_two_ranks():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29800 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-cfg3", "--cg-iters", "3", "--raster", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["cg_iterations_per_s"] > 0 and d["cg_strong_iterations_per_s"] > 0
    assert d["roofline"]["achieved"] > 0


def test_bench_runs_on_rccl_with_one_rank():
    """The RCCL ("nccl") code path of ``bench.py`` -- process group on the GPU, asynchronous object all-reduce between the
    steps, all-reduced scalars and gradients of the native CG stages -- with ONE rank (``BENCH_FORCE_DIST=1``): what a
    one-GPU box can load and run of the N > 1 job.  The multi-rank logic is the two-rank gloo test above."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    port = 29900 + (os.getpid() % 90)
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu", "--no-cfg3", "--cg-iters", "3", "--raster", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["cg_iterations_per_s"] > 0
