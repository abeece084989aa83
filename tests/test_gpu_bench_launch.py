"""`python bench.py --gpus N` starts its own rank processes (no outer launcher): two ranks share the one GPU of
the test box over gloo, and the distinct-key count of the distributed build equals that of one context over all
the reads (the ranks generate disjoint slices of the same synthetic read set)."""
import json
import os
import subprocess
import sys

import pytest

import gpu_procs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--steps", "1", "--warmup", "0", "--no-extra",
                        "--e2e-reads", "0", "--no-cpu-baseline", "--hbm-budget-gb", "4"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    return json.loads(p.stdout.decode().strip().splitlines()[-1])


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    gpu_procs.check(2)
    two = _bench("--gpus", "2", "--backend", "gloo", "--reads", "2000000", "--genome", "3000000")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    one = _bench("--gpus", "1", "--reads", "4000000", "--genome", "6000000")
    assert one["n_gpus"] == 1
    assert two["config"]["distinct_kmers"] == one["config"]["distinct_kmers"] > 0
    # value = the windows of BOTH ranks over the slowest rank's time
    windows = 2 * 2000000 * (150 - 25 + 1)
    assert abs(two["value"] * 1e6 * two["ms_per_step"] * 1e-3 / windows - 1) < 0.02


@pytest.mark.gpu
def test_c3_shape_with_four_ranks():
    """BASELINE config C3's shape -- every rank its own share of the reads of one genome, the exchange before counting
    -- with as many rank PROCESSES as a box of the pool allows on its one card (four ranks + pytest = five holders of
    six): `bench.py --gpus 4` starts them itself (gloo, the GPU shared), and the distinct-key count equals one
    context's over all the reads, for both forms of the exchange.  The eight-way shape runs in ONE process
    (tests/test_gpu_group.py: eight contexts; tests/test_gpu_parity.py: one rank routing for eight parts) and over
    eight gloo ranks on the CPU (tests/test_dist_gloo.py)."""
    gpu_procs.check(4)
    one = _bench("--gpus", "1", "--reads", "4000000", "--genome", "8000000")
    for exchange in ("records", "counted"):
        four = _bench("--gpus", "4", "--backend", "gloo", "--reads", "1000000", "--genome", "2000000", "--exchange", exchange)
        assert four["n_gpus"] == 4 and four["config"]["exchange"] == exchange
        assert four["config"]["distinct_kmers"] == one["config"]["distinct_kmers"] > 0, exchange


@pytest.mark.gpu
def test_bench_refuses_more_ranks_per_gpu_than_the_pool_allows():
    """`bench.py --backend gloo` lets ranks share a GPU; more than GOSS_BENCH_MAX_RANKS_PER_GPU (four) per card is
    refused by every rank BEFORE it opens the GPU, with a message.  Checked with the bound lowered to one and two
    ranks, so that the test itself stays small."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--reads", "1000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, GOSS_BENCH_MAX_RANKS_PER_GPU="1"))
    assert p.returncode != 0
    assert b"processes per card" in p.stderr and b"rank exit codes" in p.stderr, p.stderr.decode(errors="replace")[-1500:]
    assert not any(line.startswith(b"{") for line in p.stdout.splitlines())


def test_launcher_fails_loudly_when_a_rank_fails():
    """No GPU here: every rank exits with an error, the launcher must not hang and must exit non-zero."""
    import torch
    if torch.cuda.device_count():
        pytest.skip("a GPU is visible")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--reads", "1000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert b"no GPU visible" in p.stderr


@pytest.mark.gpu
def test_a_rank_that_dies_mid_exchange_ends_the_job():
    """Two ranks (fresh child processes of bench.py's launcher) exchange records in pieces; rank 1 dies while a piece is
    on its way (GOSS_DIST_FAIL_RANK: dist.py exits the process there).  Rank 0 then waits in a collective that will never
    complete: the launcher must notice the dead rank, end the other one and exit non-zero -- not hang."""
    gpu_procs.check(2)
    env = dict(os.environ, GOSS_DIST_TEST_HOOKS="1", GOSS_DIST_FAIL_RANK="1", GOSS_DIST_FAIL_PIECE="1", GOSS_DIST_META_GROUP="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--exchange", "records",
                        "--reads", "2000000", "--genome", "3000000", "--steps", "1", "--warmup", "0", "--no-extra", "--e2e-reads", "0",
                        "--no-cpu-baseline", "--hbm-budget-gb", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode != 0
    assert b"rank exit codes" in p.stderr and b"7" in p.stderr, p.stderr.decode(errors="replace")[-1500:]
    assert not any(line.startswith(b"{") for line in p.stdout.splitlines())          # (no bench line from a job that failed)
