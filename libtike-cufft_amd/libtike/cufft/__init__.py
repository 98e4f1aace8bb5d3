"""Alias so that ``import libtike.cufft as pt`` (the import line of every
reference script, e.g. ``/root/reference/tests/test_adjoint.py:5``) resolves to
the MI355X backend."""
from libtike.hipfft.ptycho import *  # noqa: F401,F403
from libtike.hipfft import __version__  # noqa: F401
