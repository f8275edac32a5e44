"""Packaging of the MI355X-native drop-in (mirrors the reference's setup.py:3-14: one pure-Python package, examples left out).

    pip install --no-build-isolation .        # needs hipcc (ROCm) on the build host: compiles csrc/*.hip for gfx950
    python -c "import bayesiancoresets_amd as bc"

The build step runs `make -C bayesian-coresets_amd` (the same recipe `__graft_entry__.build()` uses) and ships the
resulting libbcx.so inside the package (bayesiancoresets_amd/_lib/); a source checkout keeps using
bayesian-coresets_amd/lib/libbcx.so in-tree (bayesiancoresets_amd/_native.py looks there first)."""
import os
import shutil
import subprocess

from setuptools import setup, find_packages
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, "bayesian-coresets_amd")


class BuildWithLibrary(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", SRC, "-j8"])
        super().run()
        dst = os.path.join(self.build_lib, "bayesiancoresets_amd", "_lib")
        os.makedirs(dst, exist_ok=True)
        shutil.copy2(os.path.join(SRC, "lib", "libbcx.so"), os.path.join(dst, "libbcx.so"))
        shutil.copy2(os.path.join(ROOT, "include", "bcx.h"), os.path.join(dst, "bcx.h"))


setup(
    name="bayesiancoresets_amd",
    version="0.1.0",
    description="MI355X-native greedy sparse-NNLS coreset engine (HilbertCoreset / GIGA / Frank-Wolfe / OMP / SparseVI) "
                "behind the bayesiancoresets API",
    package_dir={"": "bayesian-coresets_amd"},
    packages=find_packages(where="bayesian-coresets_amd", exclude=("examples", "examples.*")),
    install_requires=["numpy", "scipy", "torch"],
    python_requires=">=3.8",
    cmdclass={"build_py": BuildWithLibrary},
    zip_safe=False,
    platforms="linux (ROCm, gfx950)",
)
