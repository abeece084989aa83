"""CPU-only tests: the C-ABI library loads and exports every symbol include/goss_gpu.h
declares (no compute calls without a GPU), the host C++ parsers frame records like the
reference's, and the goss command line reproduces the reference's error behaviour."""
import gzip
import os
import random
import re
import subprocess
import sys

import pytest

import gossamer_amd as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOSS = os.environ.get("GOSS_BIN") or os.path.join(ROOT, "gossamer_amd", "goss")          # (GOSS_BIN: a sanitizer build, tools/host_asan.sh)


@pytest.fixture(scope="module", autouse=True)
def built():
    if not (os.path.exists(g.binding.LIB_PATH) and os.path.exists(GOSS)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gossamer_amd", "csrc"), "all"])


def header_functions():
    text = open(os.path.join(ROOT, "include", "goss_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(goss_(?:gpu_)?[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = g.load()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(g.SYMBOLS)
    assert L.goss_gpu_abi_version() == 1
    assert L.goss_gpu_strerror(0) == b"ok"
    assert b"gfx950" in L.goss_gpu_strerror(-2)


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(g.GossGpuError) as e:
        g.Context(25)
    assert e.value.status == -2          # GOSS_ERR_NO_DEVICE: there is no CPU fallback


def test_k_range_is_checked_before_the_device():
    L = g.load()
    import ctypes as C
    h = C.c_void_p()
    assert L.goss_gpu_create(C.byref(h), 0, 64, 0, 0, None) == -6     # KmerSet::MaxK = 63
    assert L.goss_gpu_create(C.byref(h), 0, 63, 1, 0, None) == -6     # Graph::MaxK = 62
    assert L.goss_gpu_create(C.byref(h), 0, 0, 0, 0, None) == -6
    assert L.goss_gpu_create(C.byref(h), 0, 25, 7, 0, None) == -1


def test_synth_generator_is_deterministic_and_well_formed():
    a = g.synth_reads_host(500, 150, 100000, seed=1)
    b = g.synth_reads_host(500, 150, 100000, seed=1)
    assert a == b and len(a) == 500 * 151
    lines = a.split(b"\n")
    assert lines[-1] == b"" and all(len(x) == 150 for x in lines[:-1])
    assert set(a) <= set(b"ACGTN\n")
    assert sum(x.count(b"N") for x in lines) == len([r for r in range(500) if r % 97 == 96])
    # a slice of the stream equals the stream generated from that read on
    assert g.synth_reads_host(100, 150, 100000, seed=1, first_read=400) == a[400 * 151:]
    assert g.synth_reads_host(500, 150, 100000, seed=2) != a


def run_goss(*args, stdin=None):
    p = subprocess.run([GOSS] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, input=stdin, timeout=120)
    return p.returncode, p.stdout, p.stderr.decode()


def test_cli_usage_errors_match_reference_text(tmp_path):
    """App.cc:176-417 + GossOptionChecker.hh."""
    out = str(tmp_path / "x")
    rc, _, err = run_goss("build-kmer-set", "-O", out, "--bogus")
    assert rc == 1 and err == "unknown option '--bogus'\nuse\n\tgoss build-kmer-set -h\nfor more usage information.\n"
    rc, _, err = run_goss("build-kmer-set", "-O", out)
    assert rc == 1 and err.startswith("mandatory option kmer-size was not given.\nuse\n\tgoss build-kmer-set -h\n")
    rc, _, err = run_goss("build-kmer-set", "-k", "64", "-O", out)
    assert rc == 1 and "The given value of the option kmer-size was invalid.\n\tvalue 64 is out of range.\n\tthe maximum allowed value is 63.\n" in err
    rc, _, err = run_goss("build-graph", "-k", "63", "-O", out)
    assert rc == 1 and "the maximum allowed value is 62." in err
    rc, _, err = run_goss("build-kmer-set", "-k", "25")
    assert rc == 1 and "mandatory option graph-out was not given." in err
    rc, _, err = run_goss("build-kmer-set", "-k", "25", "-O", out, "-i", str(tmp_path / "missing.fq"))
    assert rc == 1 and "\tcannot open file '%s' for reading\n" % (tmp_path / "missing.fq") in err
    assert "for more usage information" not in err              # not a usage error
    rc, _, err = run_goss("build-kmer-set", "-k", "25", "-O", "/nonexistent-dir/x")
    assert rc == 1 and "\tcannot create filenames with prefix '/nonexistent-dir/x'\n" in err
    rc, _, err = run_goss("frobnicate")
    assert rc == 0 and err.startswith("unknown command 'frobnicate'\n")
    rc, out_b, _ = run_goss("--version")
    assert rc == 0 and out_b.startswith(b"goss version ")
    rc, _, err = run_goss("build-graph", "-h")
    assert rc == 1 and "--kmer-size" in err and "--log-hash-slots" not in err     # -S is build-kmer-set only
    rc, _, err = run_goss("build-kmer-set", "-h")
    assert "--log-hash-slots" in err


def _bases(oracle, payload_lines, k):
    return oracle.collect([(oracle.LINE, "l", payload_lines)], k, 0)


def test_host_parsers_frame_like_the_reference(tmp_path, oracle):
    """The byte stream the host hands to the device (one read per line) must produce the same
    k-mers, in the same order, as the reference's parsers (restated in the oracle) do."""
    rng = random.Random(2)

    def seq(n):
        return "".join(rng.choice("ACGTNacgt") for _ in range(n))

    fq_records = []
    for i in range(300):
        s = seq(rng.randint(0, 120))
        q = "".join(rng.choice("I@+#5") for _ in range(len(s)))
        if i % 7 == 0 and len(s) > 20:       # wrapped record, '@'/'+' may start a quality line
            cut = rng.randint(1, len(s) - 1)
            fq_records.append("@r%d\n%s\n%s\n+r%d\n%s\n%s\n" % (i, s[:cut], s[cut:], i, q[:cut], q[cut:]))
        elif i % 11 == 0:
            fq_records.append("@r%d\r\n%s\r\n+\r\n%s\r\n" % (i, s, q))
        else:
            fq_records.append("@r%d\n%s\n+\n%s\n" % (i, s, q))
    fq = "".join(fq_records)
    # a quality string may not *start* with '@'/'+' once it is complete; keep the generator honest
    fa = "".join(">s%d desc\n%s\n%s\n" % (i, seq(rng.randint(0, 70)), seq(rng.randint(0, 70))) for i in range(100)) + ">last\nACGTACGTAC"
    ln = "\n".join(seq(rng.randint(0, 90)) for _ in range(100))     # no trailing newline
    (tmp_path / "a.fq").write_text(fq)
    (tmp_path / "b.fa").write_text(fa)
    (tmp_path / "c.txt").write_text(ln)
    with gzip.open(tmp_path / "a.fq.gz", "wb") as f:
        f.write(fq.encode())
    for k in (5, 21):
        try:
            exp = oracle.collect([(oracle.LINE, "c", ln), (oracle.FASTA, "b", fa), (oracle.FASTQ, "a", fq)], k, 0)
        except oracle.OracleError as e:
            pytest.fail("generator produced an invalid file: %s" % e)
        for fqname in ("a.fq", "a.fq.gz"):
            rc, out, err = run_goss("dump-bases", "-i", str(tmp_path / fqname), "-I", str(tmp_path / "b.fa"),
                                    "--line-in", str(tmp_path / "c.txt"))
            assert rc == 0, err
            got = oracle.collect([(oracle.LINE, "dump", out)], k, 0)
            assert got[0] == exp[0] and got[2] == exp[2]
            assert got[1] == exp[1]             # same number of reads


def test_host_parser_errors_match_reference_text(tmp_path):
    """FastqParser.hh:89-174 / FastaParser.hh:62-68 messages, formatted as App.cc:357-363."""
    cases = [
        ("a.fq", "ACGT\n", "-i", "expected '@' at beginning of line 1"),
        ("b.fq", "@r\nACGT\n", "-i", "expected sequence data or quality header at line 3"),
        ("c.fq", "@r\nACGT\n@x\n", "-i", "expected '+' at beginning of line 3"),
        ("d.fq", "@r\nACGT\n+q\nIIII\n", "-i", "quality title does not match sequence title at line 3"),
        ("e.fq", "@r\nACGT\n+\nII\n", "-i", "length mistmatch between sequence and quality data just before line 5"),
        ("f.fa", "ACGT\n", "-I", "expected '>' at beginning of line 0"),
    ]
    for name, text, flag, msg in cases:
        p = tmp_path / name
        p.write_text(text)
        rc, _, err = run_goss("dump-bases", flag, str(p))
        assert rc == 1
        assert err == "error performing dump-bases:\n\t'%s': %s\n" % (p, msg), err
    p = tmp_path / "empty.fq"
    p.write_text("")
    rc, _, err = run_goss("dump-bases", "-i", str(p))
    assert rc == 1 and err == "error performing dump-bases:\nNo valid reads."


def test_fastqs_in_list_expansion(tmp_path, oracle):
    """GossOptionChecker::expandFilenames (GossOptionChecker.hh:405-430): one file name per
    newline-terminated line -- checked through the error path (no GPU needed)."""
    (tmp_path / "one.fq").write_text("@a\nACGTA\n+\nIIIII\n")
    (tmp_path / "two.fq").write_text("@b\nTTTTT\n+\nIIIII\n")
    (tmp_path / "three.fq").write_text("@c\nGGGGG\n+\nIIIII\n")
    # the last name has no terminating newline: the reference drops it
    (tmp_path / "list.txt").write_text("%s\n%s\n%s" % (tmp_path / "one.fq", tmp_path / "two.fq", tmp_path / "three.fq"))
    rc, out, err = run_goss("dump-bases", "-f", str(tmp_path / "list.txt"))
    assert rc == 0 and out == b"ACGTA\nTTTTT\n", err


def test_parallel_fastq_parser_matches_serial(tmp_path):
    """The multi-threaded FASTQ path (chunks parsed concurrently from guessed record boundaries,
    every guess verified against the previous chunk's end) must hand over exactly the serial
    parser's byte stream -- including files with wrapped records and quality lines that start with
    '@' or '+', where guesses fail and the parser falls back to serial framing -- and report the
    same error text with the same global line number."""
    rng = random.Random(17)

    def seq(n):
        return "".join(rng.choice("ACGTN") for _ in range(n))

    def plain(i):
        s = seq(rng.randint(20, 150))
        return "@r%d\n%s\n+\n%s\n" % (i, s, "".join(rng.choice("I5#") for _ in s))

    def nasty(i):
        s = seq(rng.randint(20, 150))
        q = "@" + "".join(rng.choice("+@I") for _ in s[1:])
        if i % 2:
            cut = len(s) // 2
            return "@r%d\n%s\n%s\n+r%d\n%s\n%s\n" % (i, s[:cut], s[cut:], i, q[:cut], q[cut:])
        return "@r%d\r\n%s\r\n+\r\n%s\r\n" % (i, s, q)

    files = {
        "plain.fq": "".join(plain(i) for i in range(6000)),
        "mixed.fq": "".join(nasty(i) if i % 50 == 0 else plain(i) for i in range(6000)),
        "nasty.fq": "".join(nasty(i) for i in range(3000)),
        "notrail.fq": "".join(plain(i) for i in range(3000)).rstrip("\n"),
    }
    env = dict(os.environ, GOSS_PARSE_CHUNK="4096")
    for name, text in files.items():
        p = tmp_path / name
        p.write_text(text, newline="")
        serial = subprocess.run([GOSS, "dump-bases", "-i", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert serial.returncode == 0, serial.stderr
        for T in ("2", "5"):
            par = subprocess.run([GOSS, "dump-bases", "-T", T, "-i", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert par.returncode == 0, par.stderr
            assert par.stdout == serial.stdout, (name, T)
        # the workers read their chunk plus some room for the record that crosses its end: with almost no room the
        # record is cut off and the chunk read again with a larger window
        par = subprocess.run([GOSS, "dump-bases", "-T", "4", "-i", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             env=dict(env, GOSS_PARSE_SLACK="7"))
        assert par.returncode == 0, par.stderr
        assert par.stdout == serial.stdout, (name, "slack")
        # the chunks framed where the file is mapped (what the workers go over to when the reads of a tmpfs file run
        # slowly: pages nobody has read yet) and by reads only, whatever the file system says
        for mode in ("1", "0"):
            for extra in ({}, {"GOSS_PARSE_SLACK": "7"}):
                par = subprocess.run([GOSS, "dump-bases", "-T", "4", "-i", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                     env=dict(env, GOSS_PARSE_MMAP=mode, **extra))
                assert par.returncode == 0, par.stderr
                assert par.stdout == serial.stdout, (name, "mapping" if mode == "1" else "reads", extra)
    # an error deep in the file: same message, same line number
    bad = files["plain.fq"].split("\n")
    bad[4 * 4000 + 2] = "-"             # the '+' line of record 4000
    p = tmp_path / "bad.fq"
    p.write_text("\n".join(bad), newline="")
    serial = subprocess.run([GOSS, "dump-bases", "-i", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    par = subprocess.run([GOSS, "dump-bases", "-T", "4", "-i", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert serial.returncode == 1 and par.returncode == 1
    assert b"expected '+' at beginning of line 16005" in serial.stderr
    assert par.stderr == serial.stderr


def test_bzip2_input_single_and_concatenated_streams(tmp_path):
    """PhysicalFileFactory.cc:262-298: "*.bz2" is read through a bzip2 filter.  The host loads the system's libbz2 at
    run time; the bytes handed to the device must be those of the plain file -- for one stream and for several
    streams one after the other (pbzip2 / cat a.bz2 b.bz2), FASTQ, FASTA and line input."""
    import bz2
    rng = random.Random(9)
    reads = ["".join(rng.choice("ACGTN") for _ in range(rng.randint(30, 160))) for _ in range(4000)]
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads))
    fa = "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads[:500]))
    (tmp_path / "a.fq").write_text(fq)
    (tmp_path / "a.fq.bz2").write_bytes(bz2.compress(fq.encode()))
    half = fq.index("@r2000\n")
    (tmp_path / "two.fq.bz2").write_bytes(bz2.compress(fq[:half].encode()) + bz2.compress(fq[half:].encode(), 1))
    (tmp_path / "b.fa").write_text(fa)
    (tmp_path / "b.fa.bz2").write_bytes(bz2.compress(fa.encode()))
    rc, plain, err = run_goss("dump-bases", "-i", str(tmp_path / "a.fq"))
    assert rc == 0, err
    for name in ("a.fq.bz2", "two.fq.bz2"):
        rc, out, err = run_goss("dump-bases", "-i", str(tmp_path / name))
        assert rc == 0, err
        assert out == plain, name
    rc, plain_fa, err = run_goss("dump-bases", "-I", str(tmp_path / "b.fa"))
    rc2, out_fa, err2 = run_goss("dump-bases", "-I", str(tmp_path / "b.fa.bz2"))
    assert rc == 0 and rc2 == 0 and out_fa == plain_fa, err + err2
    (tmp_path / "bad.fq.bz2").write_bytes(bz2.compress(fq.encode())[:-40])
    rc, _, err = run_goss("dump-bases", "-i", str(tmp_path / "bad.fq.bz2"))
    assert rc != 0 and "bzip2" in err
    # several parser threads and a compressed file larger than two parser chunks: the compressed bytes must not be
    # framed as FASTQ by the parallel parser (it maps plain files only)
    env = dict(os.environ, GOSS_PARSE_CHUNK="4096")
    for name in ("a.fq.bz2", "two.fq.bz2"):
        assert os.path.getsize(tmp_path / name) > 3 * 4096
        p = subprocess.run([GOSS, "dump-bases", "-T", "4", "-i", str(tmp_path / name)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=env, timeout=120)
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout == plain, name
    # bytes behind the last stream that are no stream: ignored with a warning, as the bzip2 tool does
    (tmp_path / "tail.fq.bz2").write_bytes(bz2.compress(fq.encode()) + b"\0" * 512)
    rc, out, err = run_goss("dump-bases", "-i", str(tmp_path / "tail.fq.bz2"))
    assert rc == 0 and out == plain and "trailing garbage" in err


def test_parallel_parser_reports_a_failed_buffer_allocation(tmp_path):
    """The buffer pool is filled lazily by a thread of its own: when it cannot get a buffer the command must end
    with the error, not wait for chunks nobody will parse."""
    rng = random.Random(3)
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r))
                 for i, r in enumerate("".join(rng.choice("ACGT") for _ in range(100)) for _ in range(40000)))
    (tmp_path / "big.fq").write_text(fq)
    env = dict(os.environ, GOSS_PARSE_CHUNK=str(1 << 20))
    ok = subprocess.run([GOSS, "dump-bases", "-T", "2", "-i", str(tmp_path / "big.fq")], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, env=env, timeout=120)
    assert ok.returncode == 0
    env["GOSS_TEST_FAIL_PARSER_ALLOC"] = "1"          # the allocator thread gets one buffer and then none
    p = subprocess.run([GOSS, "dump-bases", "-T", "2", "-i", str(tmp_path / "big.fq")], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=60)
    assert p.returncode == 1
    assert b"cannot allocate parser buffers" in p.stderr


def test_parser_threads_pack_bases_to_two_bits(tmp_path):
    """The parallel parser's workers pack their bases for goss_gpu_push_packed_host (2 bits per base + a non-base flag):
    unpacked again -- the way unpack_bases_kernel does it -- the stream is the plain one with every base in upper case
    and every non-base a newline.  Both packers (table-driven, AVX2 + BMI2 where the CPU has them) against
    gossamer_amd.binding.pack_bases, the numpy statement of the format."""
    import numpy as np
    from gossamer_amd.binding import pack_bases
    rng = random.Random(12)
    reads = ["".join(rng.choice("ACGTacgtNn.-RY") if rng.random() < 0.1 else rng.choice("ACGT") for _ in range(rng.randint(1, 300)))
             for _ in range(20000)]
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads))
    (tmp_path / "p.fq").write_text(fq)
    plain = subprocess.run([GOSS, "dump-bases", "-T", "4", "-i", str(tmp_path / "p.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, GOSS_PARSE_CHUNK="65536"), timeout=120)
    assert plain.returncode == 0, plain.stderr.decode()
    want = bytes(c if c in b"ACGT" else 10 for c in plain.stdout.upper())
    for scalar in (False, True):
        env = dict(os.environ, GOSS_PARSE_CHUNK="65536", GOSS_DUMP_PACKED="1")
        if scalar:
            env["GOSS_PACK_SCALAR"] = "1"
        p = subprocess.run([GOSS, "dump-bases", "-T", "4", "-i", str(tmp_path / "p.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=env, timeout=120)
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout == want, scalar
    # the numpy statement of the format unpacks to the same bytes
    codes, bad = pack_bases(plain.stdout)
    n = len(plain.stdout)
    idx = np.arange(n)
    c = (codes[idx // 16] >> (2 * (idx % 16)).astype(np.uint32)) & 3
    b = (bad[idx // 16] >> (idx % 16).astype(np.uint16)) & 1
    got = np.where(b == 1, 10, np.frombuffer(b"ACGT", dtype=np.uint8)[c]).astype(np.uint8).tobytes()
    assert got == want


def test_bench_children_have_time_limits():
    """bench.py's side records run as child processes with a time limit: a child that does not end is killed and
    reported, the headline line is not lost to it (a hung CLI build once took the whole bench with it)."""
    import importlib
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    t0 = time.time()
    p, err = bench.run_bounded(["sleep", "30"], 1)
    assert p is None and "timed out" in err and time.time() - t0 < 10
    p, err = bench.run_bounded(["true"], 5)
    assert err is None and p.returncode == 0


def test_tmp_dir_must_be_a_directory(tmp_path):
    """--tmp-dir (App.cc:233-240): checked before anything else is done -- a name that does not exist or is not a
    directory ends the command with the name in the message (no GPU is touched before that: runs here)."""
    (tmp_path / "file").write_text("x")
    (tmp_path / "r.txt").write_text("ACGTACGTACGTACGTACGTACGTACGTACGT\n")
    for cmd in (["build-kmer-set", "-k", "25", "--line-in", str(tmp_path / "r.txt"), "-O", str(tmp_path / "o")],
                ["merge-kmer-sets", "-G", str(tmp_path / "a"), "-G", str(tmp_path / "b"), "-O", str(tmp_path / "o")]):
        rc, out, err = run_goss(*cmd, "--tmp-dir", str(tmp_path / "file"))
        assert rc == 1 and "is not a directory" in err and str(tmp_path / "file") in err, err
        rc, out, err = run_goss(*cmd, "--tmp-dir", str(tmp_path / "nowhere"))
        assert rc == 1 and str(tmp_path / "nowhere") in err, err


def test_random_files_through_serial_parallel_and_oracle_framing():
    """tools/dbg/parser_fuzz.py: random FASTQ / FASTA files (wrapped records, '@' / '+' at the start of quality lines,
    \\r\\n, empty reads, cut-off tails, planted defects) framed serially, by the parallel framer under random thread counts /
    chunk sizes / slack / mapping, and by the oracle's restatement of FastqParser.hh:78-176 / FastaParser.hh: the same
    bases in the same order, or the same message with the same line number."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg", "parser_fuzz.py"), "120", "23"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
