"""Build hook: `pip install .` / `pip install -e .` compile the HIP extension (csrc/libgpd.so, gfx950) in-tree with hipcc
through `gym_pybullet_drones_amd._native.build()` before the Python files are collected.  The metadata is in pyproject.toml.
hipcc cross-compiles without a GPU; GPD_SKIP_NATIVE_BUILD=1 skips the step (the package then raises on first use, as always
when the library is missing: there is no CPU fallback)."""
import importlib
import os
import shutil

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gym_pybullet_drones_amd")


def _compile_native():
    # the header travels inside the package so that a wheel can re-build / be bound against without the source tree
    os.makedirs(os.path.join(PKG, "include"), exist_ok=True)
    shutil.copyfile(os.path.join(ROOT, "include", "gpd.h"), os.path.join(PKG, "include", "gpd.h"))
    if os.environ.get("GPD_SKIP_NATIVE_BUILD") == "1":
        return
    # import _native.py under a stand-in parent package: the real package's __init__ would pull torch in at build time
    import sys
    import types
    stub = types.ModuleType("_gpd_build")
    stub.__path__ = [PKG]
    sys.modules["_gpd_build"] = stub
    print("building", importlib.import_module("_gpd_build._native").build(verbose=True))


class BuildPy(build_py):
    def run(self):
        _compile_native()
        super().run()


class Develop(develop):
    def run(self):
        _compile_native()
        super().run()


setup(cmdclass={"build_py": BuildPy, "develop": Develop})
