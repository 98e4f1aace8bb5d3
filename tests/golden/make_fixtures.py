"""Convert the reference's test fixtures (DATA files of its own tests,
``/root/reference/tests/model/*``) into one
compressed ``.npz`` that can travel to the GPU box, where ``/root/reference``
does not exist.  Values are copied bit for bit (float32); nothing is computed.

    python tests/golden/make_fixtures.py

The TIFFs are 32-bit float images; ``probes_*.tiff`` hold 5 frames
(``tests/test_modes.py:31-36`` reads them as ``[nmodes,128,128]``).
"""
import os
import sys

import numpy as np
from PIL import Image

SRC = "/root/reference/tests/model"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                   "model_fixtures.npz")


def read_tiff(name):
    im = Image.open(os.path.join(SRC, name))
    frames = []
    for k in range(getattr(im, "n_frames", 1)):
        im.seek(k)
        frames.append(np.array(im, dtype=np.float32))
    return frames[0] if len(frames) == 1 else np.stack(frames)


def main():
    out = {
        "coords": np.load(os.path.join(SRC, "coords.npy")).astype(np.float32),
        "prbamp": read_tiff("prbamp.tiff"),
        "prbang": read_tiff("prbang.tiff"),
        "initpsiamp": read_tiff("initpsiamp.tiff"),
        "initpsiang": read_tiff("initpsiang.tiff"),
        "probes_amp": read_tiff("probes_amp.tiff"),
        "probes_ang": read_tiff("probes_ang.tiff"),
    }
    for k, v in out.items():
        print(k, v.dtype, v.shape, float(v.min()), float(v.max()))
    np.savez_compressed(DST, **out)
    print("wrote", DST, os.path.getsize(DST), "bytes")


if __name__ == "__main__":
    sys.exit(main())
