"""The read side on the device (goss_reader.hpp): select / rank / access of a SparseArray through its
own DenseSelect indexes, for every element, against the element list -- over all DenseSelect block
kinds and one- and two-word universes (the shapes of testSparseArray.cc / testDenseArray.cc); the
files come from the product's writer and, independently, from the oracle's."""
import random

import pytest

import gossamer_amd as g

pytestmark = pytest.mark.gpu


def _check(oracle, positions, N, M, k, corrupt=None):
    """Context with `positions` as its result; index check against the oracle-written files and the
    product-written files."""
    import torch
    words = 1 if 2 * k <= 62 else 2
    flat = []
    for p in positions:
        flat.append(p & 0xFFFFFFFFFFFFFFFF)
        if words == 2:
            flat.append(p >> 64)
    t = torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in flat], dtype=torch.int64, device="cuda")
    ones = torch.ones(len(positions), dtype=torch.int32, device="cuda")
    out = []
    with g.Context(k, g.MODE_KMER_SET, hbm_budget=512 << 20) as ctx:
        ctx.push_run(t.data_ptr(), ones.data_ptr(), len(positions))
        ctx.finish()
        mine = ctx.emit_sparse_array(t.data_ptr(), words, len(positions), N, M)
        theirs = {n[2:]: b for n, b in oracle.write_sparse_array(positions, N, M, base="sa").items()}
        assert mine == theirs
        for files in (mine, theirs):
            if corrupt:
                files = dict(files)
                corrupt(files)
            out.append(ctx.check_index(files))
    return out


def _clean(rep):
    return rep["select"] == 0 and rep["rank"] == 0 and rep["access"] == 0 and rep["failures"] == 0


def test_index_check_all_block_kinds(oracle):
    rng = random.Random(13)
    cases = []
    # uniform, dense: small blocks + the last partial block
    cases.append((sorted(rng.sample(range(1 << 26), 40000)), 1 << 26, 40000, 13))
    # bad estimate M -> small D -> sparse bitmap: intermediate / large blocks
    cases.append((sorted(rng.sample(range(1 << 40), 30000)), 1 << 40, 1 << 22, 20))
    # clustered: dense clumps separated by huge gaps
    pos = set()
    for c in range(6):
        base = rng.randrange(1 << 44)
        for _ in range(9000):
            pos.add(base + rng.randrange(1 << (8 + 3 * c)))
    pos = sorted(pos)
    cases.append((pos, 1 << 46, len(pos), 23))
    cases.append((pos, 1 << 46, 1 << 28, 23))
    for positions, N, M, k in cases:
        for rep in _check(oracle, positions, N, M, k):
            assert _clean(rep), rep


def test_index_check_wide_universes(oracle):
    rng = random.Random(14)
    for bits, n in ((72, 5000), (100, 20000), (126, 300)):
        pos = sorted({rng.getrandbits(bits) for _ in range(n)})
        for rep in _check(oracle, pos, 1 << bits, len(pos), 63):
            assert _clean(rep), (bits, rep)


def test_index_check_finds_damage(oracle):
    """A flipped bit in the low-bits column, in the high-bits bitmap and in the -d1 rank array must
    each be reported."""
    rng = random.Random(15)
    pos = sorted(rng.sample(range(1 << 30), 50000))

    def flip(name, offset, bit=0):
        def f(files):
            b = bytearray(files[name])
            b[offset % len(b)] ^= 1 << bit
            files[name] = bytes(b)
        return f

    for damage in (flip(".low-bits", 4001), flip(".high-bits", 801, 3)):
        for rep in _check(oracle, pos, 1 << 30, len(pos), 15, corrupt=damage):
            assert not _clean(rep), rep

    def bump_rank(files):
        import struct
        b = bytearray(files["-d1"])
        rank_off = struct.unpack("<Q", b[24:32])[0]          # rankArrayOffset
        v = struct.unpack("<Q", b[rank_off + 16: rank_off + 24])[0]
        b[rank_off + 16: rank_off + 24] = struct.pack("<Q", v + 1)     # third block starts one position late
        files["-d1"] = bytes(b)

    for rep in _check(oracle, pos, 1 << 30, len(pos), 15, corrupt=bump_rank):
        assert rep["select"] > 0, rep
