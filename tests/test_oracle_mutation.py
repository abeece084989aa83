"""How much of the on-disk byte layout does "pinned through the reference's reader semantics" really pin?

No reference test holds output file BYTES, so DESIGN.md section 5 pins the layout of
SparseArray / DenseSelect / WordyBitVector through the restated READERS: every answer the reference's tests check
(select, rank, access, the -d0 / -d1 selects) comes out of the written bytes.  This test measures that argument:
it flips single bytes of every data file, replays ALL answers through the oracle's readers and counts the flips that
change no answer.  A byte no answer depends on is not pinned by the readers: such bytes must be padding (the 4 096-byte
header page, alignment, the unused tail of the last word) or one of the header's statistics fields, which no reader
uses and which rest on the line-by-line restatement of DenseSelect::Builder::flush (DenseArray.cc:446-647) alone.

Two objects: a stand-alone DenseSelect over a bit vector shaped to produce every block kind that fits a test (small,
intermediate with 8/16/32-bit sub-blocks, large with 32-bit entries, the last partial block), and a SparseArray whose
-d0 (inverted sense) and -d1 files sit over its high-bits vector, with its low-bits column."""
import os
import random
import struct

import oracle_lib as o

DS_FIELDS = ["version", "flags", "indexArrayOffset", "rankArrayOffset", "logBlockSize", "blockSize", "logSampleRate",
             "sampleRate", "numBlocks", "indexSize", "smallBlocks", "smallBlocksSize", "intermediateBlocks",
             "intermediateBlocksSize", "largeBlocks", "largeBlocksSize"]
DS_STATISTICS = {"smallBlocks", "smallBlocksSize", "intermediateBlocks", "intermediateBlocksSize", "largeBlocks", "largeBlocksSize"}
DS_PAGE = 4096          # the header occupies one page; the blocks start behind it (DenseArray.cc:432-444)


def in_child(fn):
    """fn() != 0 or a crash -> True; in a child process: a flipped offset may send the reader anywhere."""
    pid = os.fork()
    if pid == 0:
        import faulthandler
        faulthandler.disable()          # a dying reader is an expected outcome here, not something to report
        code = 1
        try:
            code = 1 if fn() else 0
        except BaseException:
            code = 1
        os._exit(code)
    _, status = os.waitpid(pid, 0)
    return status != 0


def flip(files, name, off, mask):
    b = bytearray(files[name])
    b[off] ^= mask
    out = dict(files)
    out[name] = bytes(b)
    return out


def ds_header(data):
    return dict(zip(DS_FIELDS, struct.unpack_from("<16Q", data)))


def check_dense_select_file(files, name, changed, rng, nbody):
    """Header: every field by itself; body: a sample.  Returns (silent header fields, silent body offsets, sampled)."""
    data = files[name]
    silent_fields = {DS_FIELDS[i] for i in range(16) if not changed(flip(files, name, 8 * i, 0xFF))}
    offs = rng.sample(range(DS_PAGE, len(data)), min(nbody, len(data) - DS_PAGE))
    silent = [off for off in offs if not changed(flip(files, name, off, 0xFF))]
    # the page behind the header is padding by construction: nothing there is ever read
    pad = rng.sample(range(128, DS_PAGE), 6)
    assert all(not changed(flip(files, name, off, 0xFF)) for off in pad)
    return silent_fields, silent, len(offs)


def test_dense_select_every_block_kind():
    o.lib()
    rng = random.Random(5)
    ones, at = [], 1000

    def block(span, n=8192, tight_head=False):
        nonlocal at
        if tight_head:          # sub-blocks of 64 ones with different spans: 8-, 16- and 32-bit spills
            xs = sorted(rng.sample(range(200), 64)) + sorted(rng.sample(range(1000, 50000), 64)) + \
                 sorted(rng.sample(range(60000, span), n - 128))
        else:
            xs = [0] + sorted(rng.sample(range(1, span - 1), n - 2)) + [span - 1]          # exactly this span
        ones.extend(at + x for x in xs)
        at += span + 17

    block(20000)                       # small
    block(1 << 19, tight_head=True)    # intermediate, mixed sub-block forms
    block(40000)                       # small
    block((1 << 24) + 1)               # large (32-bit entries): span >= 2^24
    block(1 << 17)                     # intermediate
    block(3000, n=1000)                # the last, partial block: stored as a large one
    nbits = at + 100
    files = o.write_bits_and_select(ones, nbits, False)
    h = ds_header(files["x"])
    assert h["smallBlocks"] == 2 and h["intermediateBlocks"] == 2 and h["largeBlocks"] == 2, h

    def changed(fs):
        return in_child(lambda: o.replay_dense_select(fs, ones, nbits, False))
    assert not changed(files)
    silent_fields, silent, n = check_dense_select_file(files, "x", changed, rng, 150)
    print("DenseSelect: header fields no answer depends on: %s; %d of %d sampled block/index bytes carry no answer"
          % (sorted(silent_fields), len(silent), n))
    # the reader needs these; the statistics it does not (they are what the restatement alone pins)
    assert {"indexArrayOffset", "rankArrayOffset", "numBlocks"} <= set(DS_FIELDS) - silent_fields
    assert silent_fields <= DS_STATISTICS | {"version", "flags", "logBlockSize", "blockSize", "logSampleRate", "sampleRate", "indexSize"}
    assert len(silent) <= 0.05 * n, silent
    # the bit vector itself: every bit counts
    vbytes = (nbits + 7) // 8
    for off in rng.sample(range(vbytes), 40):
        assert changed(flip(files, "v", off, 0x10)), off


def test_sparse_array_files():
    o.lib()
    rng = random.Random(31)
    pos = set()
    for c in range(6):
        base = rng.randrange(1 << 44)
        for _ in range(3500):
            pos.add(base + rng.randrange(1 << (8 + 3 * c)))
    pos = sorted(pos)
    N = 1 << 46
    files = o.write_sparse_array(pos, N, len(pos), base="sa")
    D = struct.unpack_from("<8Q", files["sa.header"])[1]
    ones = [(p >> D) + i for i, p in enumerate(pos)]
    nbits = (N >> D) + len(pos)

    def changed(fs):
        return in_child(lambda: o.replay_sparse(fs, "sa", pos) + o.replay_sparse_highbits(fs, "sa", ones, nbits))
    assert not changed(files)
    assert ds_header(files["sa-d0"])["flags"] == 1 and ds_header(files["sa-d1"])["flags"] == 0
    for name in ("sa-d0", "sa-d1"):
        silent_fields, silent, n = check_dense_select_file(files, name, changed, rng, 50)
        print("%s: silent header fields %s; %d of %d sampled bytes silent" % (name, sorted(silent_fields), len(silent), n))
        assert {"indexArrayOffset", "rankArrayOffset", "numBlocks"} <= set(DS_FIELDS) - silent_fields
        assert len(silent) <= 0.05 * n + 1, silent
    # WordyBitVector words and the low-bits column: every sampled bit changes an answer
    for off in rng.sample(range((nbits + 7) // 8), 50):
        assert changed(flip(files, "sa.high-bits", off, 0x04)), off
    low = [n for n in files if ".low-bits" in n]
    assert low
    for name in low:
        for off in rng.sample(range(len(files[name])), 50):
            assert changed(flip(files, name, off, 0x01)), (name, off)
    # SparseArray header: D, the mask and the count change answers about the stored positions; the size is what
    # size() returns (testSparseArray.cc checks it) and bounds nothing else
    for field in (1, 3, 7):          # D, mask_lo, count
        assert changed(flip(files, "sa.header", 8 * field, 0x01)), field
    assert o.SparseReader(files, "sa").size() == N
    assert o.SparseReader(flip(files, "sa.header", 8 * 5, 0x01), "sa").size() != N
