"""BASELINE.json config C1 at its stated size: `goss build-kmer-set -k 25` and `goss build-graph -k 25` on 1 M synthetic
150 bp reads (genome 5 Mbp, seed 1) given as a 4-line FASTQ file.  The files on disk are compared byte for byte with the
oracle's build of the same FASTQ text run here, and with the md5 digests committed by tests/golden/make_golden.py
(tests/golden/c1_md5.json: the oracle's output when the fixture was made) -- the second comparison has no oracle in
the loop."""
import hashlib
import json
import os
import subprocess

import pytest

import c1_input

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOSS = os.path.join(ROOT, "gossamer_amd", "goss")


@pytest.fixture(scope="module")
def c1(tmp_path_factory):
    d = tmp_path_factory.mktemp("c1")
    fq = c1_input.fastq_bytes()
    (d / "c1.fq").write_bytes(fq)
    with open(os.path.join(ROOT, "tests", "golden", "c1_md5.json")) as f:
        golden = json.load(f)
    assert hashlib.md5(fq).hexdigest() == golden["fastq_md5"]          # the generator has not drifted
    return d, fq, golden


@pytest.mark.parametrize("cmd", ["build-kmer-set", "build-graph"])
def test_c1_files_equal_oracle_and_committed_digests(oracle, c1, cmd):
    d, fq, golden = c1
    base = "ks" if cmd == "build-kmer-set" else "gr"
    p = subprocess.run([GOSS, cmd, "-k", str(c1_input.K), "-T", "8", "-i", str(d / "c1.fq"), "-O", str(d / base), "-v"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = {n[len(base):]: (d / n).read_bytes() for n in os.listdir(d) if n.startswith(base + ".") or n.startswith(base + "-")}
    # (1) the committed digests
    want = golden[cmd]["files"]
    assert sorted(got) == sorted(want)
    for name, rec in want.items():
        assert len(got[name]) == rec["bytes"], name
        assert hashlib.md5(got[name]).hexdigest() == rec["md5"], name
    # (2) the oracle, run here on the same text
    build = oracle.build_kmer_set if cmd == "build-kmer-set" else oracle.build_graph
    exp, nwin = build([(oracle.FASTQ, "c1.fq", fq)], c1_input.K, out=base)
    assert nwin == golden[cmd]["windows"]
    assert ("k-mer windows: %d" % nwin).encode() in p.stderr or str(nwin).encode() in p.stderr
    for name, data in exp.items():
        assert got[name[len(base):]] == data, name


def test_every_parser_buffer_lent_and_none_free_does_not_hang(c1):
    """The parallel parser's consumer waits for parsed chunks in file order while the buffers of packed pushes come back
    only from inside library calls of that same thread (GossHost.cpp: "every buffer lent and none free").  Once the
    device side had got faster, one build in four hung there: all buffers out with their copies queued, the workers
    waiting for a buffer, the consumer waiting for a chunk nobody could parse.  With ONE buffer beyond the workers' and
    256 KB chunks that state is reached hundreds of times per build: twenty builds of C1's 1 M reads, each under a
    60 s limit, each with the committed digests -- and the state was met (GOSS_PARSE_STATS says how often)."""
    import re
    d, fq, golden = c1
    want = golden["build-kmer-set"]["files"]
    env = dict(os.environ, GOSS_PARSE_POOL="1", GOSS_PARSE_CHUNK=str(256 << 10), GOSS_PARSE_STATS="1")
    met = 0
    for it in range(20):
        base = "hang%d" % it
        p = subprocess.run([GOSS, "build-kmer-set", "-k", str(c1_input.K), "-T", "6", "-i", str(d / "c1.fq"), "-O", str(d / base)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60, env=env)
        assert p.returncode == 0, (it, p.stderr.decode()[-2000:])
        m = re.search(r"every buffer lent and none free: (\d+) times", p.stderr.decode())
        assert m, p.stderr.decode()[-2000:]
        met += int(m.group(1))
        for name, rec in want.items():
            data = (d / (base + name)).read_bytes()
            assert len(data) == rec["bytes"] and hashlib.md5(data).hexdigest() == rec["md5"], (it, name)
            os.unlink(d / (base + name))
    assert met > 0          # (else the pool and chunk settings no longer force the state: the test would prove nothing)
