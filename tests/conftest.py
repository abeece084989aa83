import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """The built artefacts are git-ignored: build them in-tree when a checkout lacks them (the same
    make the driver's build() runs).  A failure here is not fatal for collection -- the tests that
    need the library fail loudly on their own."""
    lib = os.path.join(ROOT, "gossamer_amd", "libgossgpu.so")
    exe = os.path.join(ROOT, "gossamer_amd", "goss")
    if not (os.path.exists(lib) and os.path.exists(exe)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "gossamer_amd", "csrc")], stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=1800, check=False)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
