"""Data adapters either side of the solver: what the reference's beamline script
does to measured data before ``run_batch`` and to the result after it
(``/root/reference/tests/catalyst/test_rec_script.py:20-102`` loader, ``:196-212`` solver inputs,
``:237-250`` writers).  Host side, NumPy only.

The reference reads one HDF5 file per view (``/data``, ``/positions_0|1``, ``/initprobe|recprobe``
plus four attributes) with ``h5py`` and writes HDF5 + TIFF stacks with ``h5py`` / ``dxchange``.  Neither
package is part of this image, so the same record is also accepted as an ``.npz`` archive with the same
member names (``from_npz`` / ``save_result_npz``); the HDF5 entry points import ``h5py`` on demand and
say so if it is missing.  TIFF stacks are written with Pillow as 32-bit float pages, one file per slice,
named like ``dxchange.write_tiff_stack`` names them (``<stem>_00000.tiff``).
"""
import os
import re

import numpy as np

__all__ = ["PtychoDataset", "solver_inputs", "save_result_npz", "save_result_h5",
           "write_tiff_stack", "read_tiff_stack"]

_ATTRS = ("detector_pixel_size", "detector_distance", "incident_wavelength", "rotation_angle")


class PtychoDataset:
    """One view: ``data`` float32 ``[nscan, ndet, ndet]`` un-fftshifted (DC at ``[0, 0]``, the layout the
    operators use), ``positions`` float32 ``[nscan, 2]`` in object pixels (``[:, 0]`` = row), ``probes``
    complex64 ``[nmodes, nprb, nprb]`` -- ``PtychoDAO`` of ``test_rec_script.py:11-19``."""

    def __init__(self, pid, data, positions, probes, rotation_angle=None):
        self.pid, self.data, self.positions = pid, data, positions
        self.probes, self.rotation_angle = probes, rotation_angle

    @classmethod
    def from_record(cls, rec, pid=0, use_original_positions=False, swap_position_axes=True,
                    reset_position_coordinates=True, use_original_probes=False, swap_probe_axes=False,
                    data_fftshift=True, view_dims=(2048, 2048), map_position_detector_pixel=1.0):
        """``rec``: mapping with ``data``, ``positions_0`` / ``positions_1``, ``initprobe`` / ``recprobe`` and
        the attributes ``detector_pixel_size, detector_distance, incident_wavelength, rotation_angle``.
        Same steps, in the same order, as ``PtychoDAO.h5_reader`` (``test_rec_script.py:40-102``)."""
        data = np.array(rec["data"], dtype=np.float32, order="C")
        if data_fftshift:                                   # :44-46 detector frames are stored centred
            data = np.fft.fftshift(data, axes=(1, 2))
        probes = np.array(rec["initprobe" if use_original_probes else "recprobe"], dtype=np.complex64, order="C")
        if swap_probe_axes:                                 # :66-69
            probes = np.array(probes.swapaxes(1, 2), order="C")
        positions = np.array(rec["positions_0" if use_original_positions else "positions_1"],
                             dtype=np.float32, order="C")
        # metres on the sample -> detector-conjugate pixels (:77-79; wavelength in Angstrom)
        pos2det = np.float64(((float(rec["detector_pixel_size"]) * probes.shape[-1])
                              / (float(rec["detector_distance"]) * 1e-10 * float(rec["incident_wavelength"])))
                             * map_position_detector_pixel)
        positions = np.float32(positions * pos2det)
        if swap_position_axes:                              # :82-84
            positions[:, (0, 1)] = positions[:, (1, 0)]
        if not reset_position_coordinates:                  # :95-96
            raise ValueError("Currently reset_position_coordinates has to be set to True.")
        positions[:, 0] -= positions[:, 0].min()            # :85-94 origin at (0, 0), keep what fits the view
        positions[:, 1] -= positions[:, 1].min()
        ids = np.where((positions[:, 1] >= 0) & (positions[:, 1] < view_dims[1])
                       & (positions[:, 0] >= 0) & (positions[:, 0] < view_dims[0]))[0]
        positions = np.array(positions[ids], dtype=np.float32, order="C")
        data = np.ascontiguousarray(data[ids])              # :98-100
        angle = rec.get("rotation_angle")                   # None for a missing key OR a missing HDF5 attribute
        return cls(pid, data, positions, probes, None if angle is None else float(angle))

    @classmethod
    def from_npz(cls, path, pid=None, **kw):
        with np.load(path) as z:
            rec = {k: z[k] for k in z.files}
        return cls.from_record(rec, _pid_of(path) if pid is None else pid, **kw)

    @classmethod
    def from_h5(cls, path, pid=None, **kw):
        """The reference's file layout (``test_rec_script.py:38-75``); needs ``h5py``."""
        try:
            import h5py
        except ImportError as e:                            # pragma: no cover - h5py is not in this image
            raise ImportError("PtychoDataset.from_h5 needs h5py; convert the file to .npz "
                              "(same member names) and use from_npz") from e
        with h5py.File(path, "r") as fid:
            rec = {k: np.array(fid[k]) for k in ("data", "positions_0", "positions_1", "initprobe", "recprobe")
                   if k in fid}
            rec.update({k: fid.attrs.get(k) for k in _ATTRS})
        return cls.from_record(rec, _pid_of(path) if pid is None else pid, **kw)


def _pid_of(path):
    """``.../extracted_scan339.h5`` -> 339 (``test_rec_script.py:34-37``: second-to-last number of the path)."""
    nums = re.findall(r"\d+", str(path))
    return int(nums[-2]) if len(nums) >= 2 else (int(nums[-1]) if nums else 0)


def solver_inputs(ds, view_dims, nmodes=1):
    """Arrays for ``CGPtychoSolver.run_batch`` from one view (``test_rec_script.py:181-212``): a leading
    angle axis of 1, a flat start object ``exp(-0.25i)`` of ``view + ndet`` pixels, the first ``nmodes``
    probes scaled to max |.| = 1 and the data scaled by the same factor squared."""
    data = ds.data[None].astype(np.float32)
    scan = ds.positions[None].astype(np.float32)
    prb = ds.probes[None, :nmodes].astype(np.complex64)
    ndet = data.shape[-1]
    psi = np.zeros((1, view_dims[0] + ndet, view_dims[1] + ndet), dtype="complex64") + np.exp(-0.25j)
    scale = np.amax(np.abs(prb))
    data = data / scale ** 2
    prb = prb / scale
    return {"data": np.ascontiguousarray(data, np.float32), "psi": psi.astype(np.complex64),
            "scan": scan, "probe": np.ascontiguousarray(prb, np.complex64)}


def save_result_npz(path, pid, psi, probe, rotation_angle=None):
    """``<pid>/psi``, ``<pid>/probe`` and the angle (``test_rec_script.py:237-244``) as an .npz archive."""
    np.savez(path, **{"%s/psi" % pid: psi, "%s/probe" % pid: probe,
                      "%s/rotation_angle" % pid: np.float32(0.0 if rotation_angle is None else rotation_angle)})


def save_result_h5(path, pid, psi, probe, rotation_angle=None):
    """The reference's result file (``test_rec_script.py:237-244``); needs ``h5py``."""
    try:
        import h5py
    except ImportError as e:                                # pragma: no cover
        raise ImportError("save_result_h5 needs h5py; use save_result_npz") from e
    with h5py.File(path, "w") as f:
        f.create_group(str(pid))
        f.create_dataset("%s/psi" % pid, data=psi)
        f.create_dataset("%s/probe" % pid, data=probe)
        f[str(pid)].attrs["rotation_angle"] = np.float32(0.0 if rotation_angle is None else rotation_angle)


def write_tiff_stack(arr, fname, overwrite=True):
    """Float32 TIFF per slice of a ``[n, h, w]`` real array, named ``<stem>_00000.tiff`` ... like
    ``dxchange.write_tiff_stack`` (``test_rec_script.py:245-250``).  Returns the file names."""
    from PIL import Image
    arr = np.asarray(arr, dtype=np.float32)
    if arr.ndim == 2:
        arr = arr[None]
    stem, ext = os.path.splitext(fname)
    os.makedirs(os.path.dirname(os.path.abspath(fname)), exist_ok=True)
    names = []
    for k, page in enumerate(arr):
        name = "%s_%05d%s" % (stem, k, ext or ".tiff")
        if os.path.exists(name) and not overwrite:
            raise FileExistsError(name)
        Image.fromarray(page, mode="F").save(name)
        names.append(name)
    return names


def read_tiff_stack(names):
    from PIL import Image
    return np.stack([np.array(Image.open(n), dtype=np.float32) for n in names])
