import sys; sys.path.insert(0,'.'); sys.path.insert(0,'libtike-cufft_amd')
import numpy as np, torch
import libtike.hipfft as pt
from oracle import ptycho_oracle as op
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
ndet = nprb = 256; ntheta = 1
rng = np.random.default_rng(9)
nscan, nz, n = 150, ndet + 90, ndet + 130
for case in ("raster", "random", "random+special"):
    scan = np.empty((ntheta, nscan, 2), np.float32)
    if case == "raster":
        yy, xx = np.meshgrid(np.arange(10) * 8.3, np.arange(15) * 8.1, indexing="ij")
        scan[0, :, 0] = yy.ravel(); scan[0, :, 1] = xx.ravel()
    else:
        scan[..., 0] = rng.random((ntheta, nscan)) * (nz - nprb - 1)
        scan[..., 1] = rng.random((ntheta, nscan)) * (n - nprb - 1)
    if case == "random+special":
        scan[0, 5] = [-3.5, 4.0]; scan[-1, 7] = [nz - nprb / 3, n - nprb / 2]; scan[0, 9:12] = scan[0, 8]
    prb = (rng.standard_normal((ntheta, nprb, nprb)) + 1j * rng.standard_normal((ntheta, nprb, nprb))).astype(np.complex64)
    y = (rng.standard_normal((ntheta, nscan, ndet, ndet)) + 1j * rng.standard_normal((ntheta, nscan, ndet, ndet))).astype(np.complex64)
    want = op.adj(y, scan, prb, nz, n, "double")
    with pt.PtychoCuFFT(nscan, nprb, ndet, ntheta, nz, n) as slv:
        for chunk in (0, 37):
            slv.set_chunk(chunk)
            got = slv.adj(dev(y), dev(scan), dev(prb)).cpu().numpy()
            d = np.abs(got - want)
            iy, ix = np.unravel_index(d[0].argmax(), d[0].shape)
            print(case, "chunk", chunk, "max rel err %.3e" % (d.max() / np.abs(want).max()), "at", iy, ix, "rows with err>1e-4:", np.unique(np.where(d[0] > 1e-4 * np.abs(want).max())[0])[:12], "cols:", np.unique(np.where(d[0] > 1e-4 * np.abs(want).max())[1])[:12])
