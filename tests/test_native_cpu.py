"""CPU-side checks of the native library: it builds for gfx950, loads, exports
every symbol ``include/ptycho_hip.h`` declares, and validates arguments before
touching the GPU.  No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libtike-cufft_amd", "csrc")


@pytest.fixture(scope="module")
def nat():
    import __graft_entry__ as ge
    ge.build_native()
    from libtike.hipfft import _native
    return _native


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptycho_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptycho_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(nat):
    names = declared_symbols()
    assert "ptycho_fwd" in names and "ptycho_adj" in names and len(names) >= 10
    for name in names:
        assert hasattr(nat.lib, name), name
    assert sorted(nat.SYMBOLS) == names
    assert b"gfx950" in nat.version()


def test_code_object_targets_gfx950(nat):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", nat.LIB_PATH],
                         capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out
    blob = open(nat.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_argument_validation_needs_no_gpu(nat):
    h = ctypes.c_void_p()
    # ndet = 1100: neither <= 1024 (Bluestein path) nor a power of two
    assert nat.create(ctypes.byref(h), 1, 1200, 1200, 4, 1100, 16) == 1
    assert b"power of two" in nat.last_error()
    assert nat.create(ctypes.byref(h), 1, 64, 64, 4, 16, 32) == 1      # nprb > ndet
    assert nat.create(ctypes.byref(h), 0, 64, 64, 4, 16, 16) == 1      # zero size
    assert h.value is None
    assert nat.fwd(None, None, None, None, None, None) == 1            # null handle
    assert nat.get(None, 0) == -1
    assert nat.destroy(None) == 0
    # every CG-stage entry point rejects a null handle before touching the device
    assert nat.cg_zoom(None, None, None, None, None, 9, 150, 100.0, None, None) == 1
    assert nat.cg_cross(None, 0, 1, 0.5, None, None) == 1
    assert nat.cg_argmax(None, 1, None, None) == 1
    assert nat.cg_linesearch(None, 0, 1, None, None, 1.0, 16, None, None) == 1
    assert b"null handle" in nat.last_error()


def test_zoom_window_kernel_is_low_rank():
    """Host side of ``ptycho_cg_zoom``: the real factors of the centred window kernel
    reproduce ``exp(i theta_k jc)`` to float64 noise with 16 terms for every supported size."""
    import numpy as np
    from libtike.hipfft import ptycho as P
    for npts in (16, 64, 256, 1024):
        vt, lz, nc = P._zoom_real_factors(npts, 150, 100, "cpu")
        vt, lz = vt.numpy(), lz.numpy()
        assert vt.shape == (npts, 16) and lz.shape == (150, 16) and 8 <= nc <= 9
        th = 2 * np.pi * np.fft.fftfreq(npts, 100)
        jc = np.arange(150) - 74.5
        want = np.exp(1j * jc[:, None] * th[None, :])
        got = lz[:, :nc] @ vt[:, :nc].T + 1j * (lz[:, nc:] @ vt[:, nc:].T)
        assert np.abs(got - want).max() < 2e-13
    assert P._zoom_real_factors(256, 150, 1, "cpu") is None      # a window that is not low rank


def test_fft_core_on_host():
    """Stockham index math of csrc/fft_core.hpp for every plan (16..2048, both
    directions) emulated thread by thread on the host against a naive DFT."""
    exe = "/tmp/pty_host_check"
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O2",
                    os.path.join(CSRC, "host_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "OK" in out.stdout.splitlines()[-1]


def test_python_api_mirrors_reference_names():
    import libtike.hipfft as pt
    import libtike.cufft as alias
    for cls in (pt.PtychoCuFFT, pt.CGPtychoSolver, alias.PtychoCuFFT, alias.CGPtychoSolver):
        for name in ("fwd", "adj", "adj_probe", "fwd_ptycho_batch", "adj_ptycho_batch",
                     "adj_ptycho_batch_prb", "run", "run_batch", "free", "__enter__", "__exit__"):
            assert hasattr(cls, name), (cls, name)
        assert hasattr(cls, "array_module") and hasattr(cls, "asnumpy")
    assert issubclass(pt.CGPtychoSolver, pt.PtychoCuFFT)
    import inspect
    sig = inspect.signature(pt.CGPtychoSolver.run)
    assert list(sig.parameters)[:9] == ["self", "data", "psi", "scan", "probe", "piter",
                                        "model", "recover_prb", "ortho_prb"]
    assert list(inspect.signature(pt.PtychoCuFFT.__init__).parameters)[1:] == \
        ["nscan", "probe_shape", "detector_shape", "ntheta", "nz", "n"]
