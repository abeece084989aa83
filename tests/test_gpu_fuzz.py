"""A short run of tests/fuzz_parity.py inside the suite: random (k, mode, read shape, error rate, arena size,
path switches) combinations, product against oracle, files byte for byte."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_against_the_oracle():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "30", "11", "groups"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-4000:]
    assert "30 cases, 0 failed" in out


def test_large_random_configurations_against_the_oracle():
    """The "large" mode: 0.4 to 1 M reads per case -- several chunks per build, the fused path taken without
    forcing, arenas of 0.5 to 8 GB."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "4", "23", "large"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-4000:]
    assert "4 cases, 0 failed" in out
