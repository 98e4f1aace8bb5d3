"""The experiments build of the library (``make -C libtike-cufft_amd/csrc experiments``, ``-DPTYCHO_EXPERIMENTS``) carries what
the shipped ``libptychohip.so`` leaves out: the environment knobs and the single-launch forward ``k_fwd_fused256`` (a
measured-slower design kept as documented negative evidence).  It is checked in a process of its
own, because the library is chosen at import time (``PTYCHO_HIP_LIB``)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "build", "libptychohip_exp.so")


@pytest.mark.gpu
def test_fused_forward_of_the_experiments_build():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(EXP):
        pytest.skip("experiments build not present (make -C libtike-cufft_amd/csrc experiments)")
    env = dict(os.environ, PTYCHO_HIP_LIB=EXP)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "exp_fused_check.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "fused forward ok" in r.stdout


@pytest.mark.gpu
def test_shipped_library_refuses_the_experiment_option():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import libtike.hipfft as pt
    from libtike.hipfft._native import PtychoHipError
    with pt.PtychoCuFFT(4, 16, 16, 1, 48, 48) as slv:
        slv.set_fused(0)
        with pytest.raises(PtychoHipError):
            slv.set_fused(2)
