"""pip install -e .  -- installs the `pnode_amd` package and the `pnode` drop-in shim, after
building pnode_amd/lib/libpnode_amd.so with hipcc for gfx950 (the library is the product; it is not
optional and there is no CPU build)."""
import os
import sys

import setuptools
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class BuildWithHip(build_py):
    def run(self):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as entry
        entry.build_library()
        super().run()


setuptools.setup(
    name="pnode_amd",
    version="0.1.0",
    description="MI355X-native neural-ODE time stepper + discrete adjoint behind pnode's ODEPetsc API",
    packages=["pnode_amd", "pnode"],
    package_data={"pnode_amd": ["lib/*.so", "csrc/*"]},
    include_package_data=True,
    install_requires=["torch"],
    cmdclass={"build_py": BuildWithHip},
)
