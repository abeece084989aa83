"""How many processes a test may have on the GPU at once.

A GPU box of the pool ends a run in which more than six processes have the card open (the pytest process counts:
it holds the GPU from its first GPU test on).  Every test that starts processes which open the GPU asks here
first; the bound -- GOSS_TEST_MAX_GPU_PROCS, four by default: four ranks + pytest = five holders -- is checked
BEFORE anything is started, so that a test written for more ranks fails by itself instead of taking the whole
suite down with it (round 5's driver run: an eight-rank test, third in collection order, ended 292 tests)."""
import os

import pytest

MAX_GPU_PROCS = int(os.environ.get("GOSS_TEST_MAX_GPU_PROCS", "4"))


def check(n, what="ranks"):
    """fail the calling test, before it starts anything, when it would put more than the bound on the card"""
    if n > MAX_GPU_PROCS:
        pytest.fail("%d %s would open the GPU beside pytest; the pool allows six holders in all "
                    "(GOSS_TEST_MAX_GPU_PROCS = %d): cover wider shapes with several contexts in ONE process "
                    "(tests/test_gpu_group.py) or with gloo on the CPU (tests/test_dist_gloo.py)" % (n, what, MAX_GPU_PROCS),
                    pytrace=False)
    return n
