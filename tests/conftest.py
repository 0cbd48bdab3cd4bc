"""pytest config: registers the ``gpu`` marker and puts the product package on sys.path."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "ct-image-segmentation_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the shared library is a build artefact (git-ignored): build it in-tree when a fresh checkout runs the tests first
    lib = os.path.join(PKG, "lib", "libctseg_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", PKG, "-j4"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load
