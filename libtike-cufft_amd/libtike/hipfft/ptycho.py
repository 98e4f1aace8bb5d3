"""Ptychography operators and CG solver on MI355X (HIP), keeping the Python
operator API of ``libtike.cufft`` (``/root/reference/src/libtike/cufft/ptycho.py``).

Device arrays are ``torch`` tensors on the current ROCm device (the reference
uses CuPy arrays); torch is used for allocation, elementwise glue and
``torch.distributed`` only -- the operators themselves are the HIP kernels behind
``libptychohip.so`` (C ABI in ``include/ptycho_hip.h``).  There is no CPU path:
importing this module without the shared library raises.

Solvers are context managers::

    with CGPtychoSolver(nscan, nprb, ndet, ptheta, nz, n) as slv:
        result = slv.run_batch(data, psi, scan, probe, piter=50)
"""
import ctypes
import warnings

import numpy as np
import torch

from . import _native as nat

__all__ = ["PtychoHIP", "PtychoCuFFT", "CGPtychoSolver",
           "register_translation_batch", "TorchArrayModule"]


class TorchArrayModule:
    """Minimal ``array_module`` hook (``ptycho.py:55`` of the reference exposes
    ``cp``): what a host framework needs to create device arrays."""
    complex64, float32, float64 = torch.complex64, torch.float32, torch.float64

    @staticmethod
    def _dev():
        return torch.device("cuda", torch.cuda.current_device())

    @classmethod
    def asarray(cls, x, dtype=None):
        if isinstance(x, torch.Tensor):
            return x.to(device=cls._dev(), dtype=dtype) if dtype else x.to(cls._dev())
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype, device=cls._dev())

    array = asarray

    @classmethod
    def zeros(cls, shape, dtype=torch.float32):
        return torch.zeros(tuple(shape), dtype=_tdtype(dtype), device=cls._dev())

    @classmethod
    def ones(cls, shape, dtype=torch.float32):
        return torch.ones(tuple(shape), dtype=_tdtype(dtype), device=cls._dev())

    @classmethod
    def empty(cls, shape, dtype=torch.float32):
        return torch.empty(tuple(shape), dtype=_tdtype(dtype), device=cls._dev())


def _tdtype(d):
    if isinstance(d, torch.dtype):
        return d
    return {"complex64": torch.complex64, "float32": torch.float32,
            "float64": torch.float64, "complex128": torch.complex128}[np.dtype(d).name]


def _asnumpy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class PtychoHIP:
    """Forward / adjoint ptychography operators (``PtychoCuFFT`` of the
    reference, ``ptycho.py:34-162``).

    Attributes
    ----------
    nscan : int   scan positions per angular view
    nprb : int    probe is ``nprb x nprb``
    ndet : int    detector is ``ndet x ndet`` (2..1024, or a power of two up to 2048; powers of two
                  run the fused kernels, other sizes a Bluestein transform)
    ptheta : int  angular views processed per call
    n, nz : int   object width, height
    """

    array_module = TorchArrayModule
    asnumpy = staticmethod(_asnumpy)

    def __init__(self, nscan, probe_shape, detector_shape, ntheta, nz, n):
        # argument order of ptycho.py:58-60 -> native (ptheta, nz, n, nscan, ndet, nprb)
        if not torch.cuda.is_available():
            raise RuntimeError("libtike.hipfft needs a ROCm GPU; there is no CPU path")
        self._h = ctypes.c_void_p()
        nat.check(nat.create(ctypes.byref(self._h), ntheta, nz, n, nscan,
                             detector_shape, probe_shape))
        self._device = torch.device("cuda", torch.cuda.current_device())
        self._det = False      # option "deterministic" as set by set_deterministic()

    # read-only size attributes of the native object (swig/ptychofft.i:11-16)
    ptheta = property(lambda self: int(nat.get(self._h, 0)))
    nz = property(lambda self: int(nat.get(self._h, 1)))
    n = property(lambda self: int(nat.get(self._h, 2)))
    nscan = property(lambda self: int(nat.get(self._h, 3)))
    ndet = property(lambda self: int(nat.get(self._h, 4)))
    nprb = property(lambda self: int(nat.get(self._h, 5)))

    def __enter__(self):
        return self

    def __exit__(self, type, value, traceback):
        self.free()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                nat.destroy(h)
            except Exception:
                pass
            self._h = None

    def free(self):
        """Release device scratch; idempotent (``ptychofft.cu:49-57``)."""
        if self._h is not None and self._h.value:
            nat.check(nat.free(self._h))

    def set_chunk(self, positions):
        """Positions per launch pair of the adjoint (0 = default: at most 4 GiB of scratch)."""
        nat.check(nat.set_option(self._h, b"chunk", int(positions)))

    def set_window(self, on=True):
        """Object adjoint: LDS overlap-add window (default) or direct atomics."""
        nat.check(nat.set_option(self._h, b"window", int(bool(on))))

    def set_split(self, on=True):
        """ndet = 256: split the DFT over y between the column and the row pass (default on)."""
        nat.check(nat.set_option(self._h, b"split", int(bool(on))))

    def set_tile(self, on=True):
        """ndet <= 128: forward operator and probe adjoint as one launch each, the tile stays in the compute unit's
        LDS (default on); off = the two-pass kernels of the larger sizes."""
        nat.check(nat.set_option(self._h, b"tile", int(bool(on))))

    def set_deterministic(self, on=True):
        """Adjoints accumulate in 64-bit fixed point (integer atomics): bitwise reproducible results
        (the reference's float ``atomicAdd``, kernels.cu:73-80,92-93, is not).  ndet <= 512."""
        nat.check(nat.set_option(self._h, b"deterministic", int(bool(on))))
        self._det = bool(on)

    def release_scratch(self):
        """Free the adjoint's intermediate (up to 4 GiB; ``adj`` allocates it again when needed): the fused CG loops never use it."""
        nat.check(nat.set_option(self._h, b"release_scratch", 1))

    def release_work(self, slot):
        """Free one farplane-sized CG work slot (the next stage that writes it allocates it again).  The native loop holds
        slots 0-3 when the position correction shares the object step's patch gathers, 0-1 (+2 with a process group) without."""
        nat.check(nat.set_option(self._h, b"release_work", int(slot)))

    def work_slots_allocated(self):
        """Indices of the CG work slots that currently hold device memory (``ptheta * nscan * ndet^2 * 8`` bytes each)."""
        return [s for s in range(16) if nat.get(self._h, 200 + s) == 1]

    def set_fused(self, tiles=2):
        """ndet = 256: forward operator as one launch (``k_fwd_fused256``), ``tiles`` = 0 (off), 1 or 2."""
        nat.check(nat.set_option(self._h, b"fused", int(tiles)))

    def profile(self, enable=True):
        """Bracket every kernel launch with HIP events (bench.py's live timing)."""
        nat.check(nat.profile(self._h, int(bool(enable))))

    def profile_read(self):
        """``{kernel: (total_ms, launches)}`` since the last read; waits for them."""
        nk = len(nat.KERNEL_NAMES)
        ms = (ctypes.c_double * nk)()
        cnt = (ctypes.c_longlong * nk)()
        nat.check(nat.profile_read(self._h, ms, cnt, nk))
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(nat.KERNEL_NAMES) if cnt[i]}

    # -- helpers -------------------------------------------------------------
    def _operand(self, x, dtype, shape, name):
        assert x.dtype == dtype, f"{name}: {x.dtype}"
        if not x.is_cuda:
            raise ValueError(f"{name} must be a device tensor")
        if tuple(x.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(x.shape)} != expected {tuple(shape)}")
        return x if x.is_contiguous() else x.contiguous()

    def _note_scan(self, scan):
        """Tell the native side whether ``scan`` is the tensor (same storage, same torch
        version counter) the previous operator call sorted; if so the sort is reused."""
        key = (scan.data_ptr(), scan._version, tuple(scan.shape))
        same = getattr(self, "_scan_key", None) == key
        if same != getattr(self, "_scan_trusted", False):
            nat.check(nat.set_option(self._h, b"trust_order", int(same)))
            self._scan_trusted = same
        self._scan_key = key

    # -- operators (ptycho.py:80-123) ---------------------------------------
    def fwd(self, psi, scan, probe, out=None):
        """Ptychography transform (FQ).  ``out``: optional farplane tensor to write into (the reference
        allocates a fresh one per call, ptycho.py:85-86)."""
        psi = self._operand(psi, torch.complex64, (self.ptheta, self.nz, self.n), "psi")
        scan = self._operand(scan, torch.float32, (self.ptheta, self.nscan, 2), "scan")
        probe = self._operand(probe, torch.complex64, (self.ptheta, self.nprb, self.nprb), "probe")
        if out is None:
            farplane = torch.empty((self.ptheta, self.nscan, self.ndet, self.ndet),
                                   dtype=torch.complex64, device=psi.device)
        else:
            farplane = self._operand(out, torch.complex64, (self.ptheta, self.nscan, self.ndet, self.ndet), "out")
            assert farplane is out, "out must be contiguous"
        self._note_scan(scan)
        nat.check(nat.fwd(self._h, _ptr(farplane), _ptr(psi), _ptr(scan), _ptr(probe), _stream()))
        return farplane

    def adj(self, farplane, scan, probe, out=None):
        """Adjoint ptychography transform (Q*F*).  ``out``: optional object tensor (zeroed here, ptycho.py:102)."""
        farplane = self._operand(farplane, torch.complex64,
                                 (self.ptheta, self.nscan, self.ndet, self.ndet), "farplane")
        scan = self._operand(scan, torch.float32, (self.ptheta, self.nscan, 2), "scan")
        probe = self._operand(probe, torch.complex64, (self.ptheta, self.nprb, self.nprb), "probe")
        if out is None:
            psi = torch.zeros((self.ptheta, self.nz, self.n), dtype=torch.complex64,
                              device=farplane.device)
        else:
            psi = self._operand(out, torch.complex64, (self.ptheta, self.nz, self.n), "out")
            assert psi is out, "out must be contiguous"
            psi.zero_()
        self._note_scan(scan)
        nat.check(nat.adj(self._h, _ptr(psi), _ptr(farplane), _ptr(scan), _ptr(probe), 0, _stream()))
        return psi

    def adj_probe(self, farplane, scan, psi):
        """Adjoint ptychography probe transform (O*F*), object is fixed."""
        farplane = self._operand(farplane, torch.complex64,
                                 (self.ptheta, self.nscan, self.ndet, self.ndet), "farplane")
        scan = self._operand(scan, torch.float32, (self.ptheta, self.nscan, 2), "scan")
        psi = self._operand(psi, torch.complex64, (self.ptheta, self.nz, self.n), "psi")
        probe = torch.zeros((self.ptheta, self.nprb, self.nprb), dtype=torch.complex64,
                            device=farplane.device)
        self._note_scan(scan)
        nat.check(nat.adj(self._h, _ptr(psi), _ptr(farplane), _ptr(scan), _ptr(probe), 1, _stream()))
        return probe

    def fft2(self, x, inverse=False, out=None):
        """Unnormalised batched 2-D DFT of ``[..., ndet, ndet]`` complex64 tiles
        (the cuFFT plan of ``ptychofft.cu:14-20``)."""
        assert x.dtype == torch.complex64 and x.shape[-1] == x.shape[-2] == self.ndet
        x = x.contiguous()
        out = torch.empty_like(x) if out is None else out
        nb = x.numel() // (self.ndet * self.ndet)
        nat.check(nat.fft2(self._h, _ptr(out), _ptr(x), nb, 1 if inverse else -1, _stream()))
        return out

    # -- host batching (ptycho.py:70-78, 91-95, 108-111, 125-129) -----------
    def _batch(self, function, output, *inputs):
        """NumPy in / NumPy out, one angular partition of ``ptheta`` views at a
        time (the reference uploads slices of length 1, which is only right for
        ``ptheta == 1``; here the slice length is ``ptheta``)."""
        xp = self.array_module
        step = self.ptheta
        for ids in range(0, inputs[0].shape[0] - step + 1, step):
            dev = [xp.asarray(x[ids:ids + step]) for x in inputs]
            # device -> final host memory in one copy (no intermediate host tensor)
            torch.from_numpy(output[ids:ids + step]).copy_(function(*dev))
        return output

    def fwd_ptycho_batch(self, psi, scan, probe):
        data = np.zeros([scan.shape[0], self.nscan, self.ndet, self.ndet], dtype="complex64")
        return self._batch(self.fwd, data, psi, scan, _single_mode(probe))

    def adj_ptycho_batch(self, farplane, scan, probe):
        psi = np.zeros([scan.shape[0], self.nz, self.n], dtype="complex64")
        return self._batch(self.adj, psi, farplane, scan, _single_mode(probe))

    def adj_ptycho_batch_prb(self, farplane, scan, psi):
        probe = np.zeros([scan.shape[0], self.nprb, self.nprb], dtype="complex64")
        return self._batch(self.adj_probe, probe, farplane, scan, psi)

    def run(self, data, psi, scan, probe, **kwargs):
        raise NotImplementedError("Cannot run a base class.")

    def run_batch(self, data, psi, scan, probe, angle_shard=None, **kwargs):
        """Run by dividing the work into angular partitions (``ptycho.py:135-162``).
        NumPy in / NumPy out; ``scan`` updates are not returned and remainder angles are
        dropped, as in the reference.

        The reference uploads, solves and downloads one partition at a time, fully
        synchronously.  Here the next partition's ``data / psi / scan / probe`` are staged
        through two reused sets of pinned buffers by a worker thread and copied on a copy
        stream while the current partition is being solved (angle streaming).  Angle partitions are independent problems, so a
        multi-GPU job gives every rank its own partitions with no collective:
        ``angle_shard=(rank, world)`` restricts this call to partitions ``rank, rank+world, ...``
        (the other entries of the returned arrays keep their input values).
        """
        assert probe.ndim == 4, "probe needs 4 dimensions, not %d" % probe.ndim
        import threading
        psi = psi.copy()
        probe = probe.copy()
        nparts = scan.shape[0] // self.ptheta
        rank, world = angle_shard if angle_shard is not None else (0, 1)
        mine = list(range(nparts))[rank::world]
        dev = self._device
        copy_stream = torch.cuda.Stream(device=dev)
        arrays = (data, psi, scan, probe)
        # two sets of pinned staging buffers, filled by a worker thread while the main thread
        # drives the solver (the pinned copy of 1 GiB of data takes longer than its DMA)
        pinned = [[torch.empty((self.ptheta,) + x.shape[1:], dtype=torch.from_numpy(x[:0]).dtype).pin_memory()
                   for x in arrays] for _ in range(min(2, len(mine)))]
        done = [None, None]                       # H2D-complete events of the two sets

        def stage(n, box):
            k = mine[n]
            ids = slice(k * self.ptheta, (k + 1) * self.ptheta)
            bufs = pinned[n % 2]
            if done[n % 2] is not None:
                done[n % 2].synchronize()         # the DMA out of this set (partition n-2) is over
            for b, x in zip(bufs, arrays):
                b.copy_(torch.from_numpy(np.ascontiguousarray(x[ids])))
            with torch.cuda.stream(copy_stream):
                t = [b.to(dev, non_blocking=True) for b in bufs]
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            done[n % 2] = ev
            box.append((ids, t, ev))

        def start(n):
            box = []
            th = threading.Thread(target=stage, args=(n, box), daemon=True)
            th.start()
            return th, box

        pending = start(0) if mine else None
        for n in range(len(mine)):
            th, box = pending
            th.join()
            ids, (d_gpu, psi_gpu, scan_gpu, prb_gpu), ev = box[0]
            torch.cuda.current_stream().wait_event(ev)
            for t in (d_gpu, psi_gpu, scan_gpu, prb_gpu):
                t.record_stream(torch.cuda.current_stream())
            pending = start(n + 1) if n + 1 < len(mine) else None
            result = self.run(d_gpu, psi_gpu, scan_gpu, prb_gpu, **kwargs)
            psi[ids] = self.asnumpy(result["psi"])
            probe[ids] = self.asnumpy(result["probe"])
        return {"psi": psi, "probe": probe}


def _single_mode(probe):
    """The ``*_batch`` wrappers accept a ``[ntheta,1,nprb,nprb]`` probe
    (``/root/reference/tests/test_adjoint.py:24,44``): the slice keeps its memory
    layout and the native side reads it as ``[ntheta,nprb,nprb]``."""
    probe = np.asarray(probe)
    if probe.ndim == 4:
        assert probe.shape[1] == 1, "the *_batch wrappers take one probe mode"
        return probe[:, 0]
    return probe


#: drop-in name of the reference class
PtychoCuFFT = PtychoHIP


def _dy_direction(i, grad, grad0, d):
    """Dai-Yuan direction with the reference's complex beta (ptycho.py:366-372)."""
    if i == 0:
        return -grad
    return -grad + (torch.linalg.norm(grad) ** 2
                    / (torch.sum(torch.conj(d) * (grad - grad0))) * d)


# ---------------------------------------------------------------------------
# position registration (ptycho.py:163-248)
# ---------------------------------------------------------------------------
_ZOOM_CACHE = {}


def _zoom_factors(npts, ups, upsample_factor, sgn, device):
    """Rank-revealing factorisation ``A = L @ R`` of the zoomed-DFT matrix
    ``A[j, k] = exp(sgn 2 pi i j f_k)``, ``j < ups``, ``f = fftfreq(npts, upsample_factor)``.

    The phase ``2 pi j f_k`` spans only a few radians over the whole matrix (the window is
    1.5 detector pixels wide), so ``A`` is numerically low rank: for 150 x 256 at
    ``upsample_factor = 100`` the singular values fall below 1e-15 of the largest after 16
    terms.  Keeping every term above 1e-16 (plus two) reproduces ``A`` to float64 rounding --
    the same error level as the summation order of a float64 GEMM -- with ~10x fewer
    multiply-adds in the two contractions."""
    key = (npts, ups, upsample_factor, sgn, str(device))
    hit = _ZOOM_CACHE.get(key)
    if hit is None:
        freq = np.fft.fftfreq(npts, upsample_factor)
        A = np.exp(sgn * 2j * np.pi * np.arange(ups)[:, None] * freq[None, :])
        u, sv, vh = np.linalg.svd(A, full_matrices=False)
        rank = min(int((sv > 1e-16 * sv[0]).sum()) + 2, len(sv))
        if rank * 2 > len(sv):                       # not low rank (tiny detectors): keep A itself
            L, R = np.eye(ups, dtype=np.complex128), A
        else:
            L, R = u[:, :rank] * sv[:rank], vh[:rank]
        hit = (torch.as_tensor(np.ascontiguousarray(L), device=device),
               torch.as_tensor(np.ascontiguousarray(R), device=device))
        _ZOOM_CACHE[key] = hit
    return hit


def _zoom_real_factors(npts, ups, upsample_factor, device, rk=16):
    """Real low-rank factors of the centred window kernel for the fused zoom kernel
    (C ABI ``ptycho_cg_zoom``): with ``th = 2 pi fftfreq(npts, upsample_factor)`` and
    ``jc = j - (ups-1)/2``, ``cos(jc th) = Lc Vc`` and ``sin(jc th) = Ls Vs``.  Returns
    ``(vt [npts, rk], lz [ups, rk], nc)`` with the cos terms in columns ``< nc``, or ``None``
    when more than ``rk`` terms are above 1e-15 of the largest singular value (the float64
    noise floor of the kernel values themselves is ~1e-14: the phase argument reaches
    hundreds of radians)."""
    key = ("real", npts, ups, upsample_factor, rk, str(device))
    if key in _ZOOM_CACHE:
        return _ZOOM_CACHE[key]
    th = 2.0 * np.pi * np.fft.fftfreq(npts, upsample_factor)
    jc = np.arange(ups) - (ups - 1) / 2.0
    arg = jc[:, None] * th[None, :]
    uc, sc, vc = np.linalg.svd(np.cos(arg), full_matrices=False)
    us, ss, vs = np.linalg.svd(np.sin(arg), full_matrices=False)
    s0 = max(sc[0], ss[0] if len(ss) else 0.0)
    ns = int((ss > 1e-15 * s0).sum())
    ncos = int((sc > 1e-15 * s0).sum())
    hit = None
    if ncos + ns <= rk and rk - ns <= len(sc):
        nc = rk - ns                                   # spare terms go to the cos part
        lz = np.concatenate([uc[:, :nc] * sc[:nc], us[:, :ns] * ss[:ns]], axis=1)
        vt = np.concatenate([vc[:nc], vs[:ns]], axis=0).T
        hit = (torch.as_tensor(np.ascontiguousarray(vt), device=device),
               torch.as_tensor(np.ascontiguousarray(lz), device=device), nc)
    _ZOOM_CACHE[key] = hit
    return hit


def _upsampled_dft_batch(data, ups, upsample_factor, axis_offsets, conj=False):
    """Two matrix-multiply DFTs on an ``ups x ups`` window (``ptycho.py:163-188``).

    The reference builds a ``[nscan, ups, ndet]`` complex128 kernel per axis,
    ``exp(-2 pi i (j - off_i) f_k)``, and contracts it with ``einsum('ijk,ipk->ijp')``.
    The kernel factors as ``A[j,k] * B[i,k]`` with ``A = exp(-2 pi i j f_k)`` shared by all
    patterns and ``B = exp(+2 pi i off_i f_k)`` a per-pattern phase, and ``A`` itself is
    numerically low rank (``_zoom_factors``: ``A = L R``).  Each contraction is therefore an
    elementwise phase multiply and a dense GEMM with the thin factor ``R``; the ``ups x ups``
    window is expanded from the small core at the end.  Same float64 math, no 2.5 GB kernel
    tensors, ~10x fewer flops.  The contractions stay torch linear algebra (SURVEY.md
    section 2, C7).

    ``conj=True`` returns ``conj(_upsampled_dft_batch(conj(data), ...))`` -- what the caller
    at ``ptycho.py:225-228`` actually needs -- by conjugating the (small) phase factors
    instead of the farplane-sized operand and result."""
    nb, nrow, ncol = data.shape
    dev = data.device
    sgn = 1.0 if conj else -1.0
    Lc, Rc = _zoom_factors(ncol, ups, upsample_factor, sgn, dev)     # columns (k)
    Lr, Rr = _zoom_factors(nrow, ups, upsample_factor, sgn, dev)     # rows (p)

    def phase(off, npts):                                            # [nb, npts]
        freq = torch.fft.fftfreq(npts, upsample_factor, dtype=torch.float64, device=dev)
        return torch.exp(-sgn * 2j * np.pi * (off[:, None] * freq[None, :]).to(torch.complex128))

    # first axis (columns, k): tmp[i, p, r] = sum_k R[r, k] B1[i, k] data[i, p, k]
    # (complex64 x complex128 promotes inside the multiply: one pass, no separate cast)
    x = torch.mul(data, phase(axis_offsets[:, 1], ncol)[:, None, :])                 # [nb, p, k] c128
    tmp = torch.matmul(x, Rc.T)                                                      # [nb, p, r]
    del x
    # second axis (rows, p): core[i, r2, r] = sum_p R[r2, p] B0[i, p] tmp[i, p, r]
    tmp.mul_(phase(axis_offsets[:, 0], nrow)[:, :, None])
    core = torch.matmul(Rr, tmp)                                                     # [nb, r2, r]
    # rec[i, j2, j] = sum L[j2, r2] core[i, r2, r] L[j, r]
    return torch.matmul(torch.matmul(Lr, core), Lc.T)                                # [nb, j2, j]


def _argmax2d(a):
    flat = a.reshape(a.shape[0], -1).argmax(1)
    w = a.shape[2]
    return torch.stack((flat // w, flat % w), dim=1)


def _zoom_shifts_native(op, image_product, best, upsample_factor):
    if image_product is None:          # the product lives in work slot 2 of the handle (ptycho_cg_cross with NULL)
        dev = best.device
        fac = _zoom_real_factors(op.ndet, int(np.ceil(upsample_factor * 1.5)), upsample_factor, dev)
        if fac is None or op.ndet % 16 or op.ndet > 1024:
            return None
        vt, lz, nc = fac
        shifts = torch.empty((best.shape[0], 2), dtype=torch.float64, device=dev)
        nat.check(nat.cg_zoom(op._h, None, _ptr(best), _ptr(vt), _ptr(lz), nc, int(np.ceil(upsample_factor * 1.5)),
                              float(upsample_factor), _ptr(shifts), _stream()))
        return shifts
    return _zoom_shifts_native_ip(op, image_product, best, upsample_factor)


def _zoom_shifts_native_ip(op, image_product, best, upsample_factor):
    """Sub-pixel stage of the registration through the fused HIP kernels
    (``ptycho_cg_zoom``): ``best`` holds the whole-pixel peaks in ``ptycho_cg_argmax``'s packed
    form (int64, low word ``0xffffffff - flat index``); returns the float64 ``[nb, 2]`` shifts
    of ``ptycho.py:209-235``, or ``None`` if the kernels do not cover this case."""
    nb, nrow, ncol = image_product.shape
    region = int(np.ceil(upsample_factor * 1.5))
    if (op is None or getattr(op, "_h", None) is None or not image_product.is_cuda
            or nrow != ncol or nrow != op.ndet or nb != op.ptheta * op.nscan
            or nrow % 16 or nrow > 1024 or region > max(256, nrow) or upsample_factor < 1
            or image_product.dtype != torch.complex64 or not image_product.is_contiguous()):
        return None
    dev = image_product.device
    fac = _zoom_real_factors(nrow, region, upsample_factor, dev)
    if fac is None:
        return None
    vt, lz, nc = fac
    shifts = torch.empty((nb, 2), dtype=torch.float64, device=dev)
    nat.check(nat.cg_zoom(op._h, _ptr(image_product), _ptr(best), _ptr(vt), _ptr(lz), nc, region,
                          float(upsample_factor), _ptr(shifts), _stream()))
    return shifts


def _finish_registration(image_product, maxima, upsample_factor, op=None):
    """Second half of ``register_translation_batch`` (``ptycho.py:209-248``): wrap the
    whole-pixel maxima, then the zoomed matrix DFT around them (fused HIP kernels when ``op``
    is given and covers the case, torch GEMMs otherwise)."""
    shape = image_product.shape
    if upsample_factor > 1 and op is not None:
        packed = 0xffffffff - (maxima[:, 0].to(torch.int64) * shape[2] + maxima[:, 1].to(torch.int64))
        shifts = _zoom_shifts_native(op, image_product, packed.contiguous(), upsample_factor)
        if shifts is not None:
            return shifts
    mid = [float(np.fix(s / 2)) for s in shape[1:]]
    shifts = maxima.to(torch.float64)
    shifts[:, 0] = torch.where(shifts[:, 0] > mid[0], shifts[:, 0] - shape[1], shifts[:, 0])
    shifts[:, 1] = torch.where(shifts[:, 1] > mid[1], shifts[:, 1] - shape[2], shifts[:, 1])
    if upsample_factor > 1:
        shifts = torch.round(shifts * upsample_factor) / upsample_factor
        region = int(np.ceil(upsample_factor * 1.5))
        dftshift = float(np.fix(region / 2.0))
        offset = dftshift - shifts * upsample_factor
        # = conj(upsampled_dft(conj(image_product))) / normalization of ptycho.py:225-229; the
        # positive normalisation does not move the arg-max and is skipped
        cross = _upsampled_dft_batch(image_product, region, upsample_factor, offset, conj=True)
        maxima = _argmax2d(torch.abs(cross)).to(torch.float64) - dftshift
        shifts = shifts + maxima / upsample_factor
    for dim in range(image_product.ndim):          # reference quirk, ptycho.py:243-245
        if shape[dim] == 1:
            shifts[dim] = 0
    return shifts


def register_translation_batch(src_image, target_image, upsample_factor=1,
                               space="real", op=None):
    """Batched sub-pixel registration by phase cross-correlation (``ptycho.py:190-248``, same
    positional signature).  ``op``: an operator whose ``fft2`` (own HIP FFT) and fused zoom kernel
    are used; without one a temporary handle for the image size is made."""
    if op is None:
        nb, ny, nx = src_image.shape
        assert ny == nx, "square images only (the detector is square, ptychofft.cuh:31)"
        with PtychoHIP(nb, nx, nx, 1, nx + 2, nx + 2) as tmp:
            return register_translation_batch(src_image, target_image, upsample_factor, space, op=tmp)
    if space.lower() == "fourier":
        src_freq, target_freq = src_image, target_image
    elif space.lower() == "real":
        src_freq = op.fft2(src_image.to(torch.complex64))
        target_freq = op.fft2(target_image.to(torch.complex64))
    shape = src_freq.shape
    image_product = src_freq * target_freq.conj()
    cross = op.fft2(image_product, inverse=True) / float(shape[1] * shape[2])
    maxima = _argmax2d(torch.abs(cross))
    return _finish_registration(image_product, maxima, upsample_factor, op=op)


# ---------------------------------------------------------------------------
# CG solver (ptycho.py:250-488)
# ---------------------------------------------------------------------------
class CGPtychoSolver(PtychoHIP):
    """Solve the ptychography problem with Dai-Yuan conjugate gradients.

    ``group``: optional ``torch.distributed`` process group; when given, the scan
    positions (``data``, ``scan``) are this rank's shard, ``psi`` / ``probe`` are
    replicated, and the object / probe gradients and every global scalar are
    all-reduced (RCCL over xGMI on MI355X).
    """

    def __init__(self, nscan, probe_shape, detector_shape, ntheta, nz, n, group=None):
        super().__init__(nscan, probe_shape, detector_shape, ntheta, nz, n)
        self.group = group
        self.history = []      # (iteration, gammapsi, gammaprb, cost) per logged iteration
        self.verbose = True
        self.log_every = 32    # the reference prints every 32 iterations (ptycho.py:475)
        self.fused = True      # gaussian loops through the fused CG-stage kernels
        self.native = True     # single-mode loop sequenced by the native stage calls (no host round trips)
        self._nscan_all = None
        self.reproducible = True  # fused CG loops use the deterministic adjoints (same trajectory every run)
        self.share_ones = True    # native loop: the position correction's column passes share the object step's patch gathers
        self.ls_two_pass = None  # native line search with few collectives (<= 16, 32, 80 step lengths); None: with a group only

    # -- distributed glue ----------------------------------------------------
    def _allreduce(self, t):
        if self.group is not None:
            import torch.distributed as dist
            if torch.is_complex(t):
                dist.all_reduce(torch.view_as_real(t), group=self.group)
            else:
                dist.all_reduce(t, group=self.group)
        return t

    def _nscan_total(self):
        """Positions over all ranks (ptycho.py:431 divides the probe gradient by nscan); one collective
        per solver, not per run."""
        if self.group is None:
            return self.nscan
        if getattr(self, "_nscan_all", None) is None:
            import torch.distributed as dist
            t = torch.tensor([float(self.nscan)], device=self._device)
            dist.all_reduce(t, group=self.group)
            self._nscan_all = int(t.item())
        return self._nscan_all

    @staticmethod
    def line_search_sqr(f, p1, p2, p3, step_length=1, step_shrink=0.5):
        """Backtracking on the closed-form quadratic (``ptycho.py:253-281``)."""
        assert step_shrink > 0 and step_shrink < 1
        m = 0
        fp1 = f(p1)
        while f(p1 + step_length ** 2 * p2 + step_length * p3) > fp1 + step_shrink * m:
            if step_length < 1e-32:
                warnings.warn("Line search failed for conjugate gradient.")
                return 0
            step_length *= step_shrink
        return step_length


    # -- fused single-mode gaussian loop -------------------------------------------------
    def _cg_fwd_cols(self, slot, obj, scan, prb):
        self._note_scan(scan)
        nat.check(nat.cg_fwd_cols(self._h, slot, _ptr(obj), _ptr(scan), _ptr(prb), _stream()))

    def _position_shifts(self, psi, dpsi, gammapsi, scan, probe):
        """Shifts of ptycho.py:398-403.  Fused form (one angle): column passes of
        fwd(psi, 1) and fwd(dpsi, 1), one row pass that forms u1 conj(u1 + gamma u2) and its
        inverse row DFT, one column pass with a fused arg-max; then the zoomed DFT."""
        ones = self.__dict__.get("_ones_probe")
        if ones is None or ones.shape != probe[:, 0].shape or ones.device != probe.device:
            ones = self._ones_probe = torch.ones_like(probe[:, 0])
        if not (self.fused and self.ptheta == 1):
            g32 = gammapsi.to(torch.float32) if isinstance(gammapsi, torch.Tensor) else gammapsi
            tmp1 = self.fwd(psi, scan, ones)[0]
            tmp2 = self.fwd(psi + g32 * dpsi, scan, ones)[0]
            return register_translation_batch(tmp1, tmp2, upsample_factor=100, space="fourier", op=self)
        self._cg_fwd_cols(0, psi, scan, ones)
        self._cg_fwd_cols(1, dpsi, scan, ones)
        # three or more probe modes (compact slot layout): the image product goes to work slot 2, which is
        # free here, instead of a farplane-sized tensor of its own
        in_slot = probe.shape[1] >= 3 and _zoom_real_factors(self.ndet, 150, 100, psi.device) is not None \
            and self.ndet % 16 == 0 and self.ndet <= 1024
        ip = None if in_slot else torch.empty((self.nscan, self.ndet, self.ndet), dtype=torch.complex64, device=psi.device)
        if isinstance(gammapsi, torch.Tensor):      # the accepted step lives on the device (float64 word)
            nat.check(nat.cg_cross_dev(self._h, 0, 1, _ptr(gammapsi), _ptr(ip) if ip is not None else None, _stream()))
        else:
            nat.check(nat.cg_cross(self._h, 0, 1, float(gammapsi), _ptr(ip) if ip is not None else None, _stream()))
        best = torch.empty(self.nscan, dtype=torch.int64, device=psi.device)
        nat.check(nat.cg_argmax(self._h, 1, _ptr(best), _stream()))
        shifts = _zoom_shifts_native(self, ip, best, 100)
        if shifts is not None:
            return shifts
        idx = 0xffffffff - (best & 0xffffffff)
        maxima = torch.stack((idx // self.ndet, idx % self.ndet), dim=1)
        return _finish_registration(ip, maxima, 100)

    def _fused_line_search(self, data, ab, costs, which="psi"):
        """All trials of ``line_search_sqr`` (ptycho.py:253-281), up to 16 step lengths per
        pass over the two work buffers (p1, p2, p3 never leave registers); returns the
        accepted step length (0 on failure).  Every step length before the accepted one is
        still evaluated and rejected, as in the reference; only the number priced per pass
        adapts: the accepted index moves slowly from one iteration to the next, so a pass
        prices two more than the last accepted index of the same search (``which``) and a
        second pass continues from there if none is accepted.  Measured alternative:
        writing the terms out once and pricing 32 steps per pass from arrays is not faster
        -- each trial step costs ~0.06 ms of sqrt/FMA work at 4096 x 256^2 wherever it is
        evaluated."""
        hints = self.__dict__.setdefault("_ls_hint", {})
        ncand = min(16, max(2, hints.get(which, 14) + 2))
        gamma0 = 1.0
        tried = 0
        while True:
            costs.zero_()
            nat.check(nat.cg_linesearch(self._h, 0, 1, _ptr(data), _ptr(ab) if ab is not None else None,
                                        gamma0, ncand, _ptr(costs), _stream()))
            self._allreduce(costs)
            c = costs.to(torch.float32).cpu().numpy()      # the reference compares float32 costs
            step = gamma0
            for j in range(ncand):
                if not (c[j] > c[ncand]):
                    hints[which] = tried + j
                    return step
                if step < 1e-32:
                    warnings.warn("Line search failed for conjugate gradient.")
                    hints[which] = 14
                    return 0
                step *= 0.5
            gamma0 = step
            tried += ncand
            ncand = 16

    # -- single-mode gaussian loop, sequenced natively ---------------------------------------
    def _native_ready(self):
        """The native stage calls cover one probe mode, any number of angles per call (the position correction touches
        angle 0 only, like ptycho.py:399-403) and detector sizes the fused zoom kernel accepts."""
        if not (self.native and self.ndet % 16 == 0 and self.ndet <= 1024):
            return None
        return _zoom_real_factors(self.ndet, 150, 100, self._device)

    def _run_native(self, data, psi, scan, probe, piter, recover_prb, zoom):
        """``CGPtychoSolver.run`` (ptycho.py:283-488), one probe mode, gaussian model.  Same kernels and
        the same arithmetic as ``_run_fused``, but every scalar of the iteration (a, b, the Dai-Yuan
        sums, the line-search costs, the accepted step lengths) stays in a float64 state vector on the
        device and the line search is decided there (C ABI ``ptycho_cg_obj_* / prb_* / ls_next``): an
        iteration is ~13 library calls and no device synchronisation; the host reads the state back only
        when it logs (every ``log_every`` iterations, as the reference prints every 32).  With a process
        group the scalar messages of a search and the two gradients are all-reduced in between."""
        dev = data.device
        data = self._operand(data, torch.float32, (self.ptheta, self.nscan, self.ndet, self.ndet), "data")
        psi = self._operand(psi, torch.complex64, (self.ptheta, self.nz, self.n), "psi").clone()
        self._operand(scan, torch.float32, (self.ptheta, self.nscan, 2), "scan")
        assert probe.dtype == torch.complex64 and probe.is_contiguous() and scan.is_contiguous()
        vt, lz, nc = zoom
        st = self.__dict__.get("_cg_state")
        if st is None or st.device != dev:
            st = self._cg_state = torch.zeros(nat.ST_WORDS, dtype=torch.float64, device=dev)
            st[nat.ST_HINT:nat.ST_HINT + 2] = 14.0
        st[nat.ST_GAMMA_PSI:nat.ST_GAMMA_PRB + 1] = 0.0       # a run without probe recovery logs step 0, as the reference prints
        h = self._h
        sp, costs = _ptr(st), st[nat.ST_COSTS:nat.ST_COSTS + nat.ST_NCOSTS]
        ones = self.__dict__.get("_ones_probe")
        if ones is None or ones.shape != probe[:, 0].shape or ones.device != dev:
            ones = self._ones_probe = torch.ones_like(probe[:, 0])
        grad, grad0, dpsi = torch.empty_like(psi), torch.zeros_like(psi), torch.zeros_like(psi)
        if recover_prb:
            gprb, gprb0, dprb = (torch.zeros_like(probe[:, 0]) for _ in range(3))
        nscan_total = float(self._nscan_total())
        dist_on = self.group is not None
        two_pass = dist_on if self.ls_two_pass is None else self.ls_two_pass
        # one GPU: nothing is all-reduced between the stages, so a line-search pass decides on its own totals (no
        # decision kernel in between) and the gradient goes from the adjoint's fixed-point image straight into the
        # Dai-Yuan pass (no fold-in pass of its own); with a process group the stages stay separate
        single = not dist_on and not two_pass
        nat.check(nat.set_option(h, b"ls_fused_decide", int(single)))
        nat.check(nat.set_option(h, b"defer_finish", int(not dist_on)))

        def line_search(which, use_ab, S):
            # one GPU: 16 + 32 + 64 more step lengths in passes that return at once when resolved; with a
            # process group every pass costs a collective: 32, then all 80 that are left (a second pass of all 112
            # would save one more collective, but a search that ends at index 30-50 -- a quarter of the bench
            # problem's iterations -- would then price 112 step lengths instead of 32: +2 ms per iteration at 4096
            # positions; ls_two_pass = "all" selects it)
            for p in ((5, 4) if two_pass == "all" else (6, 7, 4) if two_pass else (1, 2, 3) if single else (1, 2, 3, 4)):
                if dist_on:
                    self._allreduce(costs)
                nat.check(nat.cg_ls_next(h, sp, which, p, _ptr(data), use_ab, S))

        # Sharing the patch gathers keeps FOUR farplane-sized work slots on the device (0, 1 and, for the two operands of
        # the position correction, 2 and 3) instead of two: 2 x ptheta x nscan x ndet^2 x 8 bytes more (4 GiB at configs[1],
        # 16 GiB at a configs[3] shard).  Where that does not fit next to what is already allocated the loop runs without
        # sharing (two more column passes per iteration: 8.33 -> 8.39 ms at 4096 x 256^2) and gives slots 2 / 3 back.
        share_fits = self.share_ones and self.ndet <= 512 and self.ptheta == 1 and piter > 1
        if share_fits:
            slot_bytes = (self.ptheta * self.nscan + 8) * self.ndet * self.ndet * 8
            need = sum(slot_bytes for s_ in (2, 3) if nat.get(h, 200 + s_) != 1)
            if need:
                free_b = torch.cuda.mem_get_info(dev)[0]
                if free_b < need + (1 << 30):
                    torch.cuda.empty_cache()
                    free_b = torch.cuda.mem_get_info(dev)[0]
                share_fits = free_b >= need + (1 << 30)
        if not share_fits:
            for s_ in (2, 3):
                if nat.get(h, 200 + s_) == 1 and not (s_ == 2 and dist_on):   # (slot 2 serves cg_reg_prepare with a process group)
                    nat.check(nat.set_option(h, b"release_work", s_))

        def iteration(first, correct):
            """One CG iteration as a fixed sequence of launches on the current stream (no host decisions)."""
            S = _stream()
            # 1) object step (ptycho.py:325-405)
            # with the position correction on, its two operands (column passes of fwd(psi, 1) and fwd(dpsi, 1)) ride
            # along with the object step's own column passes: one patch gather per position serves both probes
            share = bool(correct) and share_fits
            # with a process group the column pass of fwd(psi, 1) is better spent under the gradient all-reduce (below)
            op_psi = _ptr(ones) if (share and not dist_on) else None
            op_dpsi = _ptr(ones) if share else None
            nat.check(nat.cg_obj_begin2(h, sp, _ptr(psi), _ptr(scan), _ptr(probe), op_psi, _ptr(data), S))
            if dist_on:
                self._allreduce(st[nat.ST_A:nat.ST_A + 2])
            nat.check(nat.cg_obj_grad(h, sp, _ptr(scan), _ptr(probe), _ptr(data), _ptr(grad), S))
            if dist_on:
                # the gradient all-reduce runs on the communicator's stream; the first operand of the position
                # correction (column pass of fwd(psi, 1): depends on psi and scan only) is computed under it
                import torch.distributed as dist
                work = dist.all_reduce(torch.view_as_real(grad), group=self.group, async_op=True)
                if correct:
                    nat.check(nat.cg_reg_prepare(h, sp, _ptr(psi), _ptr(scan), _ptr(ones), S))
                    correct = 2
                work.wait()
            nat.check(nat.cg_obj_dir2(h, sp, first, _ptr(scan), _ptr(probe), op_dpsi, _ptr(data), _ptr(grad),
                                      _ptr(grad0), _ptr(dpsi), S))
            if share:
                correct = 3
            line_search(0, 1, S)
            nat.check(nat.cg_obj_finish(h, sp, correct, _ptr(psi), _ptr(dpsi), _ptr(scan), _ptr(ones),
                                        _ptr(vt), _ptr(lz), nc, 150, 100.0, S))
            # 2) probe step (ptycho.py:409-465)
            if recover_prb:
                nat.check(nat.cg_prb_grad(h, sp, _ptr(psi), _ptr(scan), _ptr(probe), _ptr(data), _ptr(gprb), S))
                if dist_on:
                    self._allreduce(gprb)
                nat.check(nat.cg_prb_dir(h, sp, first, nscan_total, 1.0, _ptr(psi), _ptr(scan), _ptr(data),
                                         _ptr(gprb), _ptr(gprb0), _ptr(dprb), S))
                line_search(1, 0, S)
                nat.check(nat.cg_prb_finish(h, sp, _ptr(probe), _ptr(dprb), S))

        if self.verbose:
            print("# congujate gradient parameters\n"
                  "iteration, step size object, step size probe, function min")
        try:
            for i in range(piter):
                iteration(int(i == 0), int(i > 0))
                if i % self.log_every == 0:
                    snap = st[:nat.ST_LS_FAILED + 1].clone()
                    if dist_on:
                        self._allreduce(snap[nat.ST_COST:nat.ST_COST + 1])
                    snap = snap.cpu()
                    self.history.append((i, float(snap[nat.ST_GAMMA_PSI]), float(snap[nat.ST_GAMMA_PRB]),
                                         float(snap[nat.ST_COST].to(torch.float32))))
                    if self.verbose:
                        print("%4d, %.3e, %.3e, %.7e" % self.history[-1])
        finally:
            # the native loop moved scan behind torch's back: forget what the operator calls knew about it
            self._scan_key = None
            self._scan_trusted = None
            nat.check(nat.set_option(self._h, b"trust_order", 0))
            nat.check(nat.set_option(self._h, b"ls_fused_decide", 0))
            nat.check(nat.set_option(self._h, b"defer_finish", 0))
        failed = int(st[nat.ST_LS_FAILED].item())
        if failed:
            st[nat.ST_LS_FAILED] = 0.0
            for _ in range(failed):
                warnings.warn("Line search failed for conjugate gradient.")
        return {"psi": psi, "probe": probe}

    def _run_fused(self, data, psi, scan, probe, piter, recover_prb):
        """``CGPtychoSolver.run`` (ptycho.py:283-488) for one probe mode and the gaussian
        model, with every farplane-sized elementwise stage fused into the DFT row pass
        (C ABI ``ptycho_cg_*``).  Work buffer 0 holds the column pass of fwd(psi), which
        is shared by the intensity statistics, the gradient projection and the line
        search (the probe rescale ``a/b`` is linear and applied on the fly)."""
        dev = data.device
        data = self._operand(data, torch.float32, (self.ptheta, self.nscan, self.ndet, self.ndet), "data")
        psi = self._operand(psi, torch.complex64, (self.ptheta, self.nz, self.n), "psi")
        self._operand(scan, torch.float32, (self.ptheta, self.nscan, 2), "scan")
        assert probe.dtype == torch.complex64 and probe.is_contiguous()
        nscan_total = self._nscan_total()
        sums = torch.zeros(2, dtype=torch.float64, device=dev)
        cost = torch.zeros(1, dtype=torch.float64, device=dev)
        costs = torch.zeros(33, dtype=torch.float64, device=dev)
        dpsi = gradpsi0 = None
        dprb = gradprb0 = None
        gammaprb = 0
        if self.verbose:
            print("# congujate gradient parameters\n"
                  "iteration, step size object, step size probe, function min")
        for i in range(piter):
            # 1) object step ----------------------------------------------------------
            self._cg_fwd_cols(0, psi, scan, probe[:, 0])
            sums.zero_()
            nat.check(nat.cg_stats(self._h, 0, _ptr(data), _ptr(sums), _stream()))
            self._allreduce(sums)
            ab32 = sums.to(torch.float32)
            probe *= (ab32[0] / ab32[1])                                    # :344
            cost.zero_()
            nat.check(nat.cg_project(self._h, 0, 1, _ptr(data), _ptr(sums), _ptr(cost), _stream()))
            gradpsi = torch.zeros((self.ptheta, self.nz, self.n), dtype=torch.complex64, device=dev)
            nat.check(nat.cg_adj_cols(self._h, 1, _ptr(gradpsi), _ptr(scan), _ptr(probe[:, 0]), 0, _stream()))
            gradpsi /= (torch.max(torch.abs(probe[:, 0])) ** 2)
            self._allreduce(gradpsi)
            dpsi = _dy_direction(i, gradpsi, gradpsi0, dpsi)
            gradpsi0 = gradpsi
            self._cg_fwd_cols(1, dpsi, scan, probe[:, 0])
            gammapsi = 0.5 * self._fused_line_search(data, sums, costs)

            if i > 0:                                                       # :398-403
                scan[0, :] += self._position_shifts(psi, dpsi, gammapsi, scan, probe).to(scan.dtype)
            psi = psi + gammapsi * dpsi

            # 2) probe step ------------------------------------------------------------
            if recover_prb:
                if i == 0:
                    gradprb0 = probe * 0
                    dprb = probe * 0
                cost2 = torch.zeros(1, dtype=torch.float64, device=dev)
                self._cg_fwd_cols(0, psi, scan, probe[:, 0])
                nat.check(nat.cg_project(self._h, 0, 1, _ptr(data), None, _ptr(cost2), _stream()))
                g = torch.zeros((self.ptheta, self.nprb, self.nprb), dtype=torch.complex64, device=dev)
                nat.check(nat.cg_adj_cols(self._h, 1, _ptr(psi), _ptr(scan), _ptr(g), 1, _stream()))
                self._allreduce(g)
                gradprb = (g / torch.max(torch.abs(psi)) ** 2 / nscan_total * 1)[:, None]
                dprb = _dy_direction(i, gradprb, gradprb0, dprb)
                gradprb0 = gradprb
                self._cg_fwd_cols(1, psi, scan, dprb[:, 0].contiguous())
                gammaprb = 0.5 * self._fused_line_search(data, None, costs, which="prb")
                probe[:, 0] = probe[:, 0] + gammaprb * dprb[:, 0]

            if i % self.log_every == 0:
                c = cost.clone()
                self._allreduce(c)
                self.history.append((i, float(gammapsi), float(gammaprb), float(c.to(torch.float32))))
                if self.verbose:
                    print("%4d, %.3e, %.3e, %.7e" % self.history[-1])
        return {"psi": psi, "probe": probe}

    # -- fused multi-mode gaussian loop ----------------------------------------------------
    def _run_fused_multi(self, data, psi, scan, probe, piter, recover_prb):
        """``CGPtychoSolver.run`` (ptycho.py:283-488), gaussian model, 2..8 incoherent probe modes.

        Work slots (one farplane each), compact layout: slot k holds the column pass of
        fwd(psi, probe_k) -- made once per step for all modes by ONE launch that gathers the object
        patch once per position (C ABI ``ptycho_cg_fwd_cols_modes``; the reference gathers per mode,
        ptycho.py:330-333) and shared by the intensity sum, the projection and the line search (the
        probe rescale a/b is linear and applied on the fly) -- and ONE further slot M is shared by all
        modes: projected residual of one mode at a time, direction column passes.  The summed intensity
        is a float32 array written once (no per-mode farplane is ever materialised); the object line
        search, which needs fwd(dpsi, probe_k) of every mode at once, runs over M position ranges with
        the M direction column passes of a range side by side in the shared slot: M + 1 farplanes instead
        of 2 M, same work.

        Device resident since round 3: a, b, the line-search costs and the accepted step lengths stay in the float64
        state vector of the native stages; every search is enqueued in full (passes of <= 16, 16, 32, 64 step lengths,
        ``ptycho_cg_ls_obj_chunk / ls_prb_pass / ls_decide``: passes after the deciding one return at once, their column
        passes included) and the host reads the state back only when it logs."""
        dev = data.device
        M = probe.shape[1]
        data = self._operand(data, torch.float32, (self.ptheta, self.nscan, self.ndet, self.ndet), "data")
        psi = self._operand(psi, torch.complex64, (self.ptheta, self.nz, self.n), "psi")
        self._operand(scan, torch.float32, (self.ptheta, self.nscan, 2), "scan")
        assert probe.dtype == torch.complex64
        nat.check(nat.set_option(self._h, b"compact_modes", M))
        self._scan_key = None                   # the position order becomes chunk-major: sort again
        nscan_total = self._nscan_total()
        st = self.__dict__.get("_cg_state")
        if st is None or st.device != dev:
            st = self._cg_state = torch.zeros(nat.ST_WORDS, dtype=torch.float64, device=dev)
            st[nat.ST_HINT:nat.ST_HINT + 2] = 14.0
        st[nat.ST_GAMMA_PSI:nat.ST_GAMMA_PRB + 1] = 0.0
        sp = _ptr(st)
        sums = st[nat.ST_A:nat.ST_A + 2]                    # a, b (views of the state vector)
        cost = st[nat.ST_COST:nat.ST_COST + 1]
        scratch_cost = st[nat.ST_COST2:nat.ST_COST2 + 1]
        costs = st[nat.ST_COSTS:nat.ST_COSTS + nat.ST_NCOSTS]
        gpsi_w = st[nat.ST_GAMMA_PSI:nat.ST_GAMMA_PSI + 1]
        gprb_w = st[nat.ST_GAMMA_PRB:nat.ST_GAMMA_PRB + 1]
        dist_on = self.group is not None
        inten = torch.empty_like(data)
        mode = lambda arr, k: arr[:, k].contiguous()
        A = lambda k: k              # column pass of fwd(psi, probe_k)
        B = M                        # shared: residual of one mode, then column passes of fwd(direction, .)
        vpp = ctypes.c_void_p * M
        passes = (1, 2, 4, 0)        # groups of 16 step lengths the NEXT pass prices (after the hint-sized first one)

        def mode_ptrs(modes):
            keep = [mode(modes, k) for k in range(M)]
            return keep, vpp(*[t.data_ptr() for t in keep])

        def fwd_cols_all(obj, modes):           # one launch per <= 4 modes, shared patch gather
            self._note_scan(scan)
            keep, ptrs = mode_ptrs(modes)
            nat.check(nat.cg_fwd_cols_modes(self._h, M, 0, _ptr(obj), _ptr(scan), ptrs, 0, 0, _stream()))

        def sum_intensity(stats=None):          # inten = sum_k |slot A(k)|^2 (+ a, b of :342-343) in one pass
            nat.check(nat.cg_intensity_modes(self._h, M, _ptr(inten), _ptr(data),
                                             _ptr(stats) if stats is not None else None, _stream()))

        def object_line_search():
            """ptycho.py:383-393 for all modes, chunk by chunk; t1_k = (a/b) * slot A(k) (old probe), t2_k = column
            pass of fwd(dpsi, probe_k) (rescaled probe) in part k of the shared slot; 0.5 * step -> state[GAMMA_PSI]."""
            self._note_scan(scan)
            keep, ptrs = mode_ptrs(probe)
            S = _stream()
            nat.check(nat.cg_ls_begin(self._h, sp, 0, S))
            for nxt in passes:
                for c in range(M):
                    nat.check(nat.cg_ls_obj_chunk(self._h, sp, c, _ptr(dpsi), _ptr(scan), ptrs, _ptr(data), _ptr(sums), S))
                if dist_on:
                    self._allreduce(costs)
                nat.check(nat.cg_ls_decide(self._h, sp, 0, nxt, S))

        def probe_line_search(m):
            """ptycho.py:451-461: p1 = summed intensity, p2 = |fwd(psi, dprb_m)|^2, p3 = 2 Re(fwd(psi, probe_m) conj(.))."""
            S = _stream()
            nat.check(nat.cg_ls_begin(self._h, sp, 1, S))
            for nxt in passes:
                nat.check(nat.cg_ls_prb_pass(self._h, sp, m, _ptr(data), _ptr(inten), S))
                if dist_on:
                    self._allreduce(costs)
                nat.check(nat.cg_ls_decide(self._h, sp, 1, nxt, S))

        dpsi = gradpsi0 = None
        dprb = gradprb0 = gradprb = None
        if self.verbose:
            print("# congujate gradient parameters\n"
                  "iteration, step size object, step size probe, function min")
        try:
            for i in range(piter):
                # 1) object step ------------------------------------------------------------
                fwd_cols_all(psi, probe)                                            # :329-333
                sums.zero_()
                sum_intensity(sums)
                self._allreduce(sums)
                ab32 = sums.to(torch.float32)
                probe *= (ab32[0] / ab32[1])                                        # :344
                gradpsi = torch.zeros((self.ptheta, self.nz, self.n), dtype=torch.complex64, device=dev)
                cost.zero_()
                for k in range(M):                                                  # :349-356
                    pk = mode(probe, k)
                    # slot A(k) was made with the probe before its rescale: fpsi = (g s)(1/s)
                    scratch_cost.zero_()
                    nat.check(nat.cg_project_multi(self._h, A(k), B, _ptr(data), _ptr(inten), _ptr(sums), 1,
                                                   _ptr(cost if k == 0 else scratch_cost), _stream()))
                    g = torch.zeros_like(gradpsi)
                    nat.check(nat.cg_adj_cols(self._h, B, _ptr(g), _ptr(scan), _ptr(pk), 0, _stream()))
                    gradpsi += g / (torch.max(torch.abs(pk)) ** 2)
                self._allreduce(gradpsi)
                dpsi = _dy_direction(i, gradpsi, gradpsi0, dpsi)
                gradpsi0 = gradpsi
                object_line_search()                                                # :383-393 -> state[GAMMA_PSI]
                gamma32 = gpsi_w.to(torch.float32)

                if i > 0:                                                           # :398-403
                    scan[0, :] += self._position_shifts(psi, dpsi, gpsi_w, scan, probe).to(scan.dtype)
                psi = psi + gamma32 * dpsi

                # 2) probe step, one mode at a time ------------------------------------------
                if recover_prb:                                                     # :409-465
                    if i == 0:
                        gradprb = probe * 0
                        gradprb0 = probe * 0
                        dprb = probe * 0
                    for m in range(M):
                        # slots A(k) = fwd(psi, probe_k) for the current psi and probes: all of them
                        # after the object step, then only the mode that was just updated
                        if m == 0:
                            fwd_cols_all(psi, probe)
                        else:
                            self._cg_fwd_cols(A(m - 1), psi, scan, mode(probe, m - 1))
                        sum_intensity()                                             # absfprb (= p1 below)
                        scratch_cost.zero_()
                        nat.check(nat.cg_project_multi(self._h, A(m), B, _ptr(data), _ptr(inten), None, 0,
                                                       _ptr(scratch_cost), _stream()))
                        g = torch.zeros((self.ptheta, self.nprb, self.nprb), dtype=torch.complex64, device=dev)
                        nat.check(nat.cg_adj_cols(self._h, B, _ptr(psi), _ptr(scan), _ptr(g), 1, _stream()))
                        self._allreduce(g)
                        gradprb[:, m] = g / torch.max(torch.abs(psi)) ** 2 / nscan_total * M
                        if i == 0:
                            dprb[:, m] = -gradprb[:, m]
                        else:
                            dprb[:, m] = -gradprb[:, m] + (
                                torch.linalg.norm(gradprb[:, m]) ** 2
                                / (torch.sum(torch.conj(dprb[:, m]) * (gradprb[:, m] - gradprb0[:, m])))
                                * dprb[:, m])
                        gradprb0[:, m] = gradprb[:, m]
                        self._cg_fwd_cols(B, psi, scan, mode(dprb, m))
                        probe_line_search(m)                                        # -> state[GAMMA_PRB]
                        probe[:, m] = probe[:, m] + gprb_w.to(torch.float32) * dprb[:, m]

                if i % self.log_every == 0:
                    snap = st[:nat.ST_LS_FAILED + 1].clone()
                    if dist_on:
                        self._allreduce(snap[nat.ST_COST:nat.ST_COST + 1])
                    snap = snap.cpu()
                    self.history.append((i, float(snap[nat.ST_GAMMA_PSI]), float(snap[nat.ST_GAMMA_PRB]),
                                         float(snap[nat.ST_COST].to(torch.float32))))
                    if self.verbose:
                        print("%4d, %.3e, %.3e, %.7e" % self.history[-1])
        finally:
            nat.check(nat.set_option(self._h, b"compact_modes", 0))
            self._scan_key = None
        failed = int(st[nat.ST_LS_FAILED].item())
        if failed:
            st[nat.ST_LS_FAILED] = 0.0
            for _ in range(failed):
                warnings.warn("Line search failed for conjugate gradient.")
        return {"psi": psi, "probe": probe}

    def run(self, data, psi, scan, probe, piter, model="gaussian",
            recover_prb=False, ortho_prb=False):
        """Conjugate gradients for ptychography (``ptycho.py:283-488``).

        ``probe`` and ``scan`` are updated in place, like in the reference.
        """
        assert probe.ndim == 4, "probe needs 4 dimensions, not %d" % probe.ndim
        nmodes = probe.shape[1]
        # the fused CG stages run on the detector sizes that have a Stockham plan of their own (csrc/fft_core.hpp): powers
        # of two and 48, 80, 96, 112, 192 (112 = the reference's own crop, tests/test_fsc.py:115-120); any other size: Bluestein
        # operators + the statement-by-statement loop below
        pow2 = (self.ndet >= 16 and (self.ndet & (self.ndet - 1)) == 0) or self.ndet in (48, 80, 96, 112, 192)
        # several modes: the compact slot layout runs its line search over position ranges, which needs the windowed
        # column pass (ndet <= 512); larger detectors take the statement-by-statement loop
        if self.fused and model == "gaussian" and pow2 and nmodes <= 8 and (nmodes == 1 or self.ndet <= 512):
            # The fused loops run on the deterministic adjoints unless told otherwise: with float atomics (the
            # reference's kernels.cu:73-80) two runs of the same problem take different line-search paths -- near a
            # flat start the accept / reject decisions sit on the last float32 digit of the cost -- and differ by
            # +-10 % in time (tools/cg_variance.py).  In the loop the fixed-point scale comes from the projection
            # stage, so this costs no extra pass.
            det = self.reproducible and not self._det and self.ndet <= 512 and int(nat.get(self._h, 101)) == 1
            if det:
                nat.check(nat.set_option(self._h, b"deterministic", 1))
            try:
                if nmodes == 1:
                    zoom = self._native_ready()
                    if zoom is not None:
                        return self._run_native(data, psi, scan, probe, piter, recover_prb, zoom)
                    return self._run_fused(data, psi, scan, probe, piter, recover_prb)
                return self._run_fused_multi(data, psi, scan, probe, piter, recover_prb)   # one pair of work slots per mode
            finally:
                if det:
                    nat.check(nat.set_option(self._h, b"deterministic", 0))
        nscan_total = self._nscan_total()

        def minf(fpsi):
            if model == "gaussian":
                f = torch.sum((torch.sqrt(torch.abs(fpsi)) - torch.sqrt(data)) ** 2)
            elif model == "poisson":
                f = torch.sum(torch.abs(fpsi) - data * torch.log(torch.abs(fpsi) + 1e-32))
            return self._allreduce(f)

        def intensity(obj):
            acc = torch.zeros_like(data)
            for k in range(nmodes):
                acc += torch.abs(self.fwd(obj, scan, probe[:, k])) ** 2
            return acc

        dprb = dpsi = gradprb0 = gradpsi0 = 0
        if self.verbose:
            print("# congujate gradient parameters\n"
                  "iteration, step size object, step size probe, function min")
        gammaprb = 0
        for i in range(piter):
            # 1) object retrieval subproblem with fixed probes -- :325-405
            absfpsi = intensity(psi)
            ab = torch.stack((torch.sum(torch.sqrt(absfpsi * data)), torch.sum(absfpsi)))
            self._allreduce(ab)
            a, b = ab[0], ab[1]
            probe *= (a / b)
            absfpsi *= (a / b) ** 2
            gradpsi = torch.zeros((self.ptheta, self.nz, self.n), dtype=torch.complex64,
                                  device=data.device)
            if model == "gaussian":
                for k in range(nmodes):
                    fpsi = self.fwd(psi, scan, probe[:, k]) * (b / a)
                    gradpsi += self.adj(
                        fpsi - torch.sqrt(data) * fpsi / (torch.sqrt(absfpsi) + 1e-32),
                        scan, probe[:, k]) / (torch.max(torch.abs(probe[:, k])) ** 2)
            elif model == "poisson":
                for k in range(nmodes):
                    gradpsi += self.adj(
                        fpsi - data * fpsi / (absfpsi + 1e-32),    # noqa: F821 (reference bug kept)
                        scan, probe[:, k]) / (torch.max(torch.abs(probe[:, k])) ** 2)
            self._allreduce(gradpsi)
            # Dai-Yuan direction
            if i == 0:
                dpsi = -gradpsi
            else:
                dpsi = -gradpsi + (
                    torch.linalg.norm(gradpsi) ** 2
                    / (torch.sum(torch.conj(dpsi) * (gradpsi - gradpsi0))) * dpsi)
            gradpsi0 = gradpsi
            p1, p2, p3 = torch.zeros_like(data), torch.zeros_like(data), torch.zeros_like(data)
            for k in range(nmodes):
                tmp1 = self.fwd(psi, scan, probe[:, k])
                tmp2 = self.fwd(dpsi, scan, probe[:, k])
                p1 += torch.abs(tmp1) ** 2
                p2 += torch.abs(tmp2) ** 2
                p3 += 2 * (tmp1.real * tmp2.real + tmp1.imag * tmp2.imag)
            gammapsi = 0.5 * self.line_search_sqr(minf, p1, p2, p3)

            # position correction -- :398-403
            if i > 0:
                ones = probe[:, 0] * 0 + 1
                tmp1 = self.fwd(psi, scan, ones)[0]
                tmp2 = self.fwd(psi + gammapsi * dpsi, scan, ones)[0]
                shifts = register_translation_batch(tmp1, tmp2, upsample_factor=100,
                                                    space="fourier", op=self)
                scan[0, :] += shifts.to(scan.dtype)
            psi = psi + gammapsi * dpsi

            if recover_prb:                     # :409-465
                if i == 0:
                    gradprb = probe * 0
                    gradprb0 = probe * 0
                    dprb = probe * 0
                for m in range(nmodes):
                    fprb = self.fwd(psi, scan, probe[:, m])
                    absfprb = intensity(psi)
                    if model == "gaussian":
                        g = self.adj_probe(
                            fprb - torch.sqrt(data) * fprb / (torch.sqrt(absfprb) + 1e-32),
                            scan, psi)
                        self._allreduce(g)
                        gradprb[:, m] = g / torch.max(torch.abs(psi)) ** 2 / nscan_total * nmodes
                    elif model == "poisson":
                        g = self.adj_probe(fprb - data * fprb / (absfprb + 1e-32), scan, psi)
                        self._allreduce(g)
                        gradprb[:, m] = g / torch.max(torch.abs(psi)) ** 2 / nscan_total
                    if i == 0:
                        dprb[:, m] = -gradprb[:, m]
                    else:
                        dprb[:, m] = -gradprb[:, m] + (
                            torch.linalg.norm(gradprb[:, m]) ** 2
                            / (torch.sum(torch.conj(dprb[:, m]) * (gradprb[:, m] - gradprb0[:, m])))
                            * dprb[:, m])
                    gradprb0[:, m] = gradprb[:, m]
                    p1 = intensity(psi)
                    tmp1 = self.fwd(psi, scan, probe[:, m])
                    tmp2 = self.fwd(psi, scan, dprb[:, m])
                    p2 = torch.abs(tmp2) ** 2
                    p3 = 2 * (tmp1.real * tmp2.real + tmp1.imag * tmp2.imag)
                    gammaprb = 0.5 * self.line_search_sqr(minf, p1, p2, p3, step_length=1)
                    probe[:, m] = probe[:, m] + gammaprb * dprb[:, m]

            # check convergence -- :475-482 (cost of the start-of-iteration intensity)
            if i % self.log_every == 0:
                cost = float(minf(absfpsi))
                self.history.append((i, float(gammapsi), float(gammaprb), cost))
                if self.verbose:
                    print("%4d, %.3e, %.3e, %.7e" % self.history[-1])
        return {"psi": psi, "probe": probe}
