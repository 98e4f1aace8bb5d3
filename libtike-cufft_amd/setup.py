"""Packaging of the MI355X backend.  Mirrors /root/reference/setup.py:1-32: namespace
package ``libtike`` + a ``tike.PtychoBackend`` entry point, but the native part is a
plain shared library built by ``make -C csrc`` (hipcc, gfx950) instead of a
scikit-build/CMake CUDA extension."""
import subprocess
import os
from setuptools import setup, find_namespace_packages
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildWithNative(build_py):
    def run(self):
        subprocess.run(["make", "-C", os.path.join(HERE, "csrc")], check=True)
        super().run()


setup(
    name="libtike-hipfft",
    version="0.1.0",
    packages=find_namespace_packages(include=["libtike.*"]),
    package_data={"libtike.hipfft": ["libptychohip.so"]},
    cmdclass={"build_py": BuildWithNative},
    zip_safe=False,
    entry_points={
        "tike.PtychoBackend": [
            "hipfft = libtike.hipfft.ptycho:PtychoHIP",
        ],
    },
)
