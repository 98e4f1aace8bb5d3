"""``libtike.hipfft`` -- MI355X (gfx950) backend with the ``libtike.cufft``
operator API (``/root/reference/src/libtike/cufft/__init__.py:1-9``)."""
from libtike.hipfft.ptycho import *  # noqa: F401,F403

__version__ = "0.1.0"
