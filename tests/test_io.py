"""Host-side data adapters (libtike.hipfft.io) against the steps of the reference's beamline loader
(/root/reference/tests/catalyst/test_rec_script.py:20-102, 196-212, 237-250)."""
import numpy as np
import pytest

from libtike.hipfft import io


def record(rng, nscan=12, ndet=16, nmodes=3):
    data = rng.random((nscan, ndet, ndet)).astype(np.float32)
    pos_m = np.stack([rng.uniform(-2e-6, 2e-6, nscan), rng.uniform(-1e-6, 3e-6, nscan)], axis=1)   # metres, (x, y)
    return {"data": data, "positions_0": pos_m * 0.9, "positions_1": pos_m,
            "initprobe": (rng.random((nmodes, ndet, ndet)) + 0j).astype(np.complex64),
            "recprobe": (rng.random((nmodes, ndet, ndet)) * np.exp(1j * rng.random((nmodes, ndet, ndet)))).astype(np.complex64),
            "detector_pixel_size": 75e-6, "detector_distance": 2.0, "incident_wavelength": 1.4,
            "rotation_angle": 12.5}


def test_loader_steps(tmp_path):
    rng = np.random.default_rng(0)
    rec = record(rng)
    path = tmp_path / "extracted_scan339_v2.npz"
    np.savez(path, **rec)
    ds = io.PtychoDataset.from_npz(path, view_dims=(64, 64))
    assert ds.pid == 339 and ds.rotation_angle == 12.5
    # metres -> pixels (test_rec_script.py:77-79), columns swapped (:82-84), origin reset (:85-87)
    k = (75e-6 * 16) / (2.0 * 1e-10 * 1.4)
    want = np.float32(rec["positions_1"] * k)[:, ::-1].copy()
    want -= want.min(axis=0)
    keep = np.where((want[:, 0] < 64) & (want[:, 1] < 64))[0]
    np.testing.assert_allclose(ds.positions, want[keep], rtol=1e-5, atol=1e-5)
    assert ds.positions.dtype == np.float32 and ds.positions.flags["C_CONTIGUOUS"]
    # frames: centred on disk, DC at [0, 0] in memory (:44-46), and only the kept positions (:98-100)
    np.testing.assert_array_equal(ds.data, np.fft.fftshift(rec["data"], axes=(1, 2))[keep])
    np.testing.assert_array_equal(ds.probes, rec["recprobe"])
    ds0 = io.PtychoDataset.from_record(rec, 7, use_original_positions=True, use_original_probes=True,
                                       swap_probe_axes=True, swap_position_axes=False, data_fftshift=False,
                                       view_dims=(4096, 4096))
    np.testing.assert_array_equal(ds0.probes, rec["initprobe"].swapaxes(1, 2))
    np.testing.assert_array_equal(ds0.data, rec["data"])
    assert ds0.positions.shape == (12, 2) and ds0.positions.min() == 0
    with pytest.raises(ValueError):
        io.PtychoDataset.from_record(rec, 7, reset_position_coordinates=False)


def test_solver_inputs_and_writers(tmp_path):
    rng = np.random.default_rng(1)
    ds = io.PtychoDataset.from_record(record(rng), 5, view_dims=(64, 64))
    inp = io.solver_inputs(ds, (64, 64), nmodes=2)
    assert inp["psi"].shape == (1, 80, 80) and np.allclose(inp["psi"], np.exp(-0.25j))
    assert inp["probe"].shape == (1, 2, 16, 16) and np.isclose(np.abs(inp["probe"]).max(), 1.0)
    s = np.abs(ds.probes[:2]).max()
    np.testing.assert_allclose(inp["data"][0], ds.data / s ** 2, rtol=1e-6)      # :205-206
    assert inp["scan"].shape == (1, ds.positions.shape[0], 2)
    # every position keeps its patch inside the start object (positions < view, object = view + ndet)
    assert (inp["scan"][0] + 16 + 1 <= 80).all()
    names = io.write_tiff_stack(np.angle(inp["probe"][0]), str(tmp_path / "probe_angle" / "rec.tiff"))
    assert [n.split("/")[-1] for n in names] == ["rec_00000.tiff", "rec_00001.tiff"]
    np.testing.assert_array_equal(io.read_tiff_stack(names), np.angle(inp["probe"][0]).astype(np.float32))
    io.save_result_npz(tmp_path / "res.npz", ds.pid, inp["psi"], inp["probe"], ds.rotation_angle)
    with np.load(tmp_path / "res.npz") as z:
        np.testing.assert_array_equal(z["5/psi"], inp["psi"])
        assert float(z["5/rotation_angle"]) == 12.5
    with pytest.raises(FileExistsError):
        io.write_tiff_stack(np.zeros((1, 4, 4)), str(tmp_path / "probe_angle" / "rec.tiff"), overwrite=False)


@pytest.mark.gpu
def test_record_to_reconstruction(tmp_path):
    """A simulated beamline record (centred frames, positions in metres) through the loader and the solver."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import libtike.hipfft as pt
    from libtike.hipfft import synthetic as syn
    from oracle import ptycho_oracle as op
    ndet, view = 32, (48, 48)
    p = syn.make_problem(5, 5, 7, ndet, ndet, seed=3, nz=view[0] + ndet, n=view[1] + ndet)
    rng = np.random.default_rng(2)
    probe = (p["probe"] * np.exp(2j * np.pi * rng.random((ndet, ndet)))).astype(np.complex64)
    frames = np.abs(op.fwd(p["psi"], p["scan"], probe, ndet)) ** 2
    k = (75e-6 * ndet) / (2.0 * 1e-10 * 1.4)
    shift = p["scan"][0] - p["scan"][0].min(axis=0)                      # the loader resets the origin
    rec = {"data": np.fft.ifftshift(frames[0], axes=(1, 2)).astype(np.float32), "positions_1": (shift / k)[:, ::-1],
           "positions_0": (shift / k)[:, ::-1], "recprobe": probe * 3.0, "initprobe": probe,
           "detector_pixel_size": 75e-6, "detector_distance": 2.0, "incident_wavelength": 1.4, "rotation_angle": 0.0}
    np.savez(tmp_path / "scan_1_x.npz", **rec)
    ds = io.PtychoDataset.from_npz(tmp_path / "scan_1_x.npz", view_dims=view)
    inp = io.solver_inputs(ds, view)
    assert inp["data"].shape == (1, 25, ndet, ndet)
    with pt.CGPtychoSolver(25, ndet, ndet, 1, view[0] + ndet, view[1] + ndet) as slv:
        slv.verbose, slv.log_every = False, 1
        res = slv.run_batch(inp["data"], inp["psi"], inp["scan"], inp["probe"], piter=8)
        cost = [h[3] for h in slv.history]
    assert cost[-1] < 0.2 * cost[0] and np.all(np.diff(cost) <= 1e-6 * cost[0])
    io.save_result_npz(tmp_path / "out.npz", ds.pid, res["psi"], res["probe"], ds.rotation_angle)
