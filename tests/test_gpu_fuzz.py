"""A short run of tests/fuzz_parity.py inside the suite: random (k, mode, read shape, error rate, arena size,
path switches -- the 32-bit-remainder forms with their second / third level bits and tables, the first level's key
space --, records of both widths, groups of contexts with the exchange before or after counting, the command line with
--devices) combinations, product against oracle, files byte for byte."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_against_the_oracle():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "28", "11", "groups"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-4000:]
    assert "28 cases, 0 failed" in out


def test_large_random_configurations_against_the_oracle():
    """The "large" mode: 0.4 to 1 M reads per case -- several chunks per build, the fused path taken without
    forcing, arenas of 0.5 to 8 GB.  (Six cases here, twenty-eight of the small kind above: the suite has 600 of the
    driver's 900 seconds; outside the suite the same script has run thousands.)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "6", "23", "large"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=2400)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-4000:]
    assert "6 cases, 0 failed" in out


def test_random_merges_and_set_algebra_against_the_oracle():
    """tests/fuzz_setops.py: objects built by `goss` from random overlapping read sets (empty ones and duplicates among
    them), then merge-kmer-sets / merge-graphs with random --max-merge and --tmp-dir, intersect / subtract /
    merge-and-annotate, graph-to-kmer-set, dump and restore -- every output file against the oracle's."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_setops.py"), "5", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-4000:]
    assert "5 cases, 0 failed" in out
