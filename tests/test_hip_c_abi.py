"""The drop-in boundary is a plain C ABI: a C++ client with no Python / torch in the loop
(tests/c_abi/abi_smoke.cpp) builds against include/ptycho_hip.h, links libptychohip.so and
passes the adjoint identities and an analytic known answer on a real GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libtike-cufft_amd", "libtike", "hipfft")
EXE = os.path.join(ROOT, "tests", "c_abi", "abi_smoke")


def build():
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "abi_smoke.cpp"), "-L", LIBDIR, "-lptychohip",
                    "-Wl,-rpath," + LIBDIR, "-o", EXE], check=True)


def test_c_client_builds_and_links():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_client_runs_on_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(EXE):
        build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI smoke OK" in out.stdout
