"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under gossamer_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")

LINE, FASTA, FASTQ = 0, 1, 2


class Key(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]

    def __int__(self):
        return (self.hi << 64) | self.lo


def key(v):
    return Key(v & 0xFFFFFFFFFFFFFFFF, v >> 64)


class Input(C.Structure):
    _fields_ = [("kind", C.c_int), ("name", C.c_char_p), ("data", C.c_char_p), ("size", C.c_size_t)]


class Keys(C.Structure):
    _fields_ = [("keys", C.POINTER(Key)), ("n", C.c_size_t), ("cap", C.c_size_t),
                ("nreads", C.c_uint64), ("nwindows", C.c_uint64)]


def build():
    if os.environ.get("GOSS_ORACLE_SO"):          # (a sanitizer build of the oracle, say: tools/oracle_asan.sh)
        return os.environ["GOSS_ORACLE_SO"]
    src = os.path.join(_ROOT, "oracle", "goss_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.go_rev64.restype = C.c_uint64
    L.go_rev64.argtypes = [C.c_uint64]
    L.go_revcomp.restype = Key
    L.go_revcomp.argtypes = [Key, C.c_uint]
    L.go_hash.restype = C.c_uint64
    L.go_hash.argtypes = [Key]
    L.go_normalize.restype = Key
    L.go_normalize.argtypes = [Key, C.c_uint]
    L.go_select1.restype = C.c_uint64
    L.go_select1.argtypes = [C.c_uint64, C.c_uint64]
    L.go_log2.restype = C.c_uint64
    L.go_log2.argtypes = [C.c_uint64]
    L.go_kmerize.restype = C.c_size_t
    L.go_kmerize.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.POINTER(Key), C.c_size_t]
    L.go_fs_new.restype = C.c_void_p
    L.go_fs_free.argtypes = [C.c_void_p]
    L.go_fs_count.restype = C.c_size_t
    L.go_fs_count.argtypes = [C.c_void_p]
    L.go_fs_name.restype = C.c_char_p
    L.go_fs_name.argtypes = [C.c_void_p, C.c_size_t]
    L.go_fs_size.restype = C.c_size_t
    L.go_fs_size.argtypes = [C.c_void_p, C.c_size_t]
    L.go_fs_data.restype = C.POINTER(C.c_uint8)
    L.go_fs_data.argtypes = [C.c_void_p, C.c_size_t]
    L.go_fs_add.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.go_collect.argtypes = [C.POINTER(Input), C.c_size_t, C.c_uint, C.c_int, C.POINTER(Keys), C.c_char_p, C.c_size_t]
    L.go_keys_free.argtypes = [C.POINTER(Keys)]
    L.go_sort_count.restype = C.c_size_t
    L.go_sort_count.argtypes = [C.POINTER(Key), C.c_size_t, C.POINTER(C.c_uint64)]
    L.go_sparse_d.restype = C.c_uint64
    L.go_sparse_d.argtypes = [Key, C.c_uint64]
    L.go_write_kmer_set.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.POINTER(Key), C.c_size_t, C.c_uint64]
    L.go_write_graph.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.POINTER(Key), C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64]
    L.go_write_sparse_array.argtypes = [C.c_void_p, C.c_char_p, Key, C.c_uint64, C.POINTER(Key), C.c_size_t, Key]
    for f in (L.go_build_kmer_set, L.go_build_graph):
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.POINTER(Input), C.c_size_t, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    L.go_sparse_open.restype = C.c_void_p
    L.go_sparse_open.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.go_sparse_close.argtypes = [C.c_void_p]
    L.go_sparse_count.restype = C.c_uint64
    L.go_sparse_count.argtypes = [C.c_void_p]
    L.go_sparse_size.restype = Key
    L.go_sparse_size.argtypes = [C.c_void_p]
    L.go_sparse_select.restype = Key
    L.go_sparse_select.argtypes = [C.c_void_p, C.c_uint64]
    L.go_sparse_rank.restype = C.c_uint64
    L.go_sparse_rank.argtypes = [C.c_void_p, Key]
    L.go_sparse_access.restype = C.c_int
    L.go_sparse_access.argtypes = [C.c_void_p, Key]
    L.go_sparse_d0_select.restype = C.c_uint64
    L.go_sparse_d0_select.argtypes = [C.c_void_p, C.c_uint64]
    L.go_sparse_d1_select.restype = C.c_uint64
    L.go_sparse_d1_select.argtypes = [C.c_void_p, C.c_uint64]
    L.go_vba_get.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]
    L.go_kmer_set_header.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.go_graph_header.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.go_vbyte_encode.restype = C.c_size_t
    L.go_vbyte_encode.argtypes = [C.c_uint64, C.POINTER(C.c_uint8)]
    L.go_vbyte_decode.restype = C.c_uint64
    L.go_vbyte_decode.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)]
    _lib = L
    return L


class OracleError(RuntimeError):
    pass


# --------------------------------------------------------------------------------------
# conveniences
# --------------------------------------------------------------------------------------

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def kmer_value(s):
    v = 0
    for c in s:
        v = (v << 2) | _CODE[c.upper()]
    return v


def kmer_string(v, k):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp(v, k):
    return int(lib().go_revcomp(key(v), k))


def fnv(v):
    return lib().go_hash(key(v))


def normalize(v, k):
    return int(lib().go_normalize(key(v), k))


def kmerize(seq, k):
    if isinstance(seq, str):
        seq = seq.encode()
    cap = max(1, len(seq))
    buf = (Key * cap)()
    n = lib().go_kmerize(seq, len(seq), k, buf, cap)
    return [int(buf[i]) for i in range(n)]


def _inputs(inputs):
    arr = (Input * len(inputs))()
    keep = []
    for i, (kind, name, data) in enumerate(inputs):
        if isinstance(data, str):
            data = data.encode()
        keep.append(data)
        arr[i] = Input(kind, name.encode(), data, len(data))
    return arr, keep


def collect(inputs, length, mode):
    """Key stream of the reference's adapters (mode 0 canonical k-mers, 1 both strands)."""
    arr, keep = _inputs(inputs)
    ks = Keys()
    err = C.create_string_buffer(512)
    rc = lib().go_collect(arr, len(inputs), length, mode, C.byref(ks), err, 512)
    if rc:
        lib().go_keys_free(C.byref(ks))
        raise OracleError(err.value.decode())
    out = [int(ks.keys[i]) for i in range(ks.n)]
    nreads, nwin = ks.nreads, ks.nwindows
    lib().go_keys_free(C.byref(ks))
    return out, nreads, nwin


def count(inputs, length, mode):
    """(sorted distinct keys, their counts, reads, windows) of the reference's key stream: go_collect + go_sort_count,
    the keys converted through numpy -- what the tests used to do with a Python dict over every window's key, which
    took most of the GPU suite's time."""
    import numpy as np
    arr, keep = _inputs(inputs)
    ks = Keys()
    err = C.create_string_buffer(512)
    rc = lib().go_collect(arr, len(inputs), length, mode, C.byref(ks), err, 512)
    if rc:
        lib().go_keys_free(C.byref(ks))
        raise OracleError(err.value.decode())
    n = ks.n
    nreads, nwin = ks.nreads, ks.nwindows
    if n == 0:
        lib().go_keys_free(C.byref(ks))
        return [], [], nreads, nwin
    counts = (C.c_uint64 * n)()
    m = lib().go_sort_count(ks.keys, n, counts)
    raw = np.ctypeslib.as_array(C.cast(ks.keys, C.POINTER(C.c_uint64)), shape=(n, 2))[:m].copy()
    cnt = np.ctypeslib.as_array(counts)[:m].copy()
    lib().go_keys_free(C.byref(ks))
    lo, hi = raw[:, 0].tolist(), raw[:, 1].tolist()
    keys = lo if not any(hi) else [(h << 64) | l for l, h in zip(lo, hi)]
    return keys, cnt.tolist(), nreads, nwin


class FileSet:
    """In-memory output file set (role of the reference's StringFileFactory)."""

    def __init__(self):
        self._fs = lib().go_fs_new()

    def __del__(self):
        if self._fs:
            lib().go_fs_free(self._fs)
            self._fs = None

    @property
    def handle(self):
        return self._fs

    def files(self):
        L = lib()
        out = {}
        for i in range(L.go_fs_count(self._fs)):
            n = L.go_fs_size(self._fs, i)
            p = L.go_fs_data(self._fs, i)
            out[L.go_fs_name(self._fs, i).decode()] = bytes(C.string_at(p, n)) if n else b""
        return out

    def add(self, name, data):
        lib().go_fs_add(self._fs, name.encode(), data, len(data))

    @classmethod
    def from_files(cls, files):
        fs = cls()
        for k, v in files.items():
            fs.add(k, v)
        return fs


def build_kmer_set(inputs, K, out="ks"):
    arr, keep = _inputs(inputs)
    fs = FileSet()
    err = C.create_string_buffer(512)
    nwin = C.c_uint64(0)
    rc = lib().go_build_kmer_set(fs.handle, out.encode(), K, arr, len(inputs), C.byref(nwin), err, 512)
    if rc:
        raise OracleError(err.value.decode())
    return fs.files(), nwin.value


def build_graph(inputs, K, out="gr"):
    arr, keep = _inputs(inputs)
    fs = FileSet()
    err = C.create_string_buffer(512)
    nwin = C.c_uint64(0)
    rc = lib().go_build_graph(fs.handle, out.encode(), K, arr, len(inputs), C.byref(nwin), err, 512)
    if rc:
        raise OracleError(err.value.decode())
    return fs.files(), nwin.value


def write_kmer_set(keys, K, M=None, out="ks"):
    n = len(keys)
    arr = (Key * max(1, n))(*[key(v) for v in keys])
    fs = FileSet()
    rc = lib().go_write_kmer_set(fs.handle, out.encode(), K, arr, n, n if M is None else M)
    if rc:
        raise OracleError("go_write_kmer_set rc=%d" % rc)
    return fs.files()


def write_graph(keys, counts, K, M=None, out="gr"):
    n = len(keys)
    arr = (Key * max(1, n))(*[key(v) for v in keys])
    cs = (C.c_uint64 * max(1, n))(*counts)
    fs = FileSet()
    rc = lib().go_write_graph(fs.handle, out.encode(), K, arr, cs, n, n if M is None else M)
    if rc:
        raise OracleError("go_write_graph rc=%d" % rc)
    return fs.files()


def write_sparse_array(positions, N, M, base="sa", N_end=None):
    n = len(positions)
    arr = (Key * max(1, n))(*[key(v) for v in positions])
    fs = FileSet()
    rc = lib().go_write_sparse_array(fs.handle, base.encode(), key(N), M, arr, n, key(N if N_end is None else N_end))
    if rc:
        raise OracleError("go_write_sparse_array rc=%d" % rc)
    return fs.files()


class SparseReader:
    """The reference's SparseArray read side, restated (select / rank / access)."""

    def __init__(self, files, base):
        self._fsobj = FileSet.from_files(files)
        err = C.create_string_buffer(512)
        self._h = lib().go_sparse_open(self._fsobj.handle, base.encode(), err, 512)
        if not self._h:
            raise OracleError(err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().go_sparse_close(self._h)
            self._h = None

    def count(self):
        return lib().go_sparse_count(self._h)

    def size(self):
        return int(lib().go_sparse_size(self._h))

    def select(self, r):
        return int(lib().go_sparse_select(self._h, r))

    def rank(self, pos):
        return lib().go_sparse_rank(self._h, key(pos))

    def access(self, pos):
        return bool(lib().go_sparse_access(self._h, key(pos)))

    def d0_select(self, i):
        return lib().go_sparse_d0_select(self._h, i)

    def d1_select(self, i):
        return lib().go_sparse_d1_select(self._h, i)


def vba_get(files, base, i):
    fs = FileSet.from_files(files)
    out = C.c_uint32(0)
    err = C.create_string_buffer(512)
    if lib().go_vba_get(fs.handle, base.encode(), i, C.byref(out), err, 512):
        raise OracleError(err.value.decode())
    return out.value


def kmer_set_header(files, base):
    fs = FileSet.from_files(files)
    K, n = C.c_uint64(0), C.c_uint64(0)
    rc = lib().go_kmer_set_header(fs.handle, base.encode(), C.byref(K), C.byref(n))
    if rc:
        raise OracleError("bad KmerSet header rc=%d" % rc)
    return K.value, n.value


def graph_header(files, base):
    fs = FileSet.from_files(files)
    K, fl = C.c_uint64(0), C.c_uint64(0)
    rc = lib().go_graph_header(fs.handle, base.encode(), C.byref(K), C.byref(fl))
    if rc:
        raise OracleError("bad Graph header rc=%d" % rc)
    return K.value, fl.value


def vbyte_encode(x):
    buf = (C.c_uint8 * 16)()
    n = lib().go_vbyte_encode(x, buf)
    return bytes(buf[:n])


def vbyte_decode(b):
    buf = (C.c_uint8 * (len(b) + 9))(*b)
    used = C.c_size_t(0)
    v = lib().go_vbyte_decode(buf, C.byref(used))
    return v, used.value


def merge(files, names, kind, out, max_merge=8):
    """merge-kmer-sets (kind 0) / merge-graphs (kind 1) over objects held in `files`."""
    L = lib()
    L.go_merge.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.c_uint64, C.c_void_p, C.c_char_p,
                           C.c_char_p, C.c_size_t]
    src = FileSet.from_files(files)
    dst = FileSet()
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    err = C.create_string_buffer(1024)
    rc = L.go_merge(src.handle, arr, len(names), kind, max_merge, dst.handle, out.encode(), err, 1024)
    if rc:
        raise OracleError(err.value.decode())
    return dst.files()


def dump(files, name, kind):
    """dump-kmer-set (kind 0) / dump-graph (kind 1) text."""
    L = lib()
    L.go_dump.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    src = FileSet.from_files(files)
    text, n = C.c_void_p(), C.c_size_t()
    err = C.create_string_buffer(1024)
    if L.go_dump(src.handle, name.encode(), kind, C.byref(text), C.byref(n), err, 1024):
        raise OracleError(err.value.decode())
    out = C.string_at(text, n.value)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(text)
    return out


def restore_graph(text, out):
    L = lib()
    L.go_restore_graph.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    dst = FileSet()
    err = C.create_string_buffer(1024)
    if L.go_restore_graph(text, len(text), dst.handle, out.encode(), err, 1024):
        raise OracleError(err.value.decode())
    return dst.files()


def intersect_kmer_sets(files, names, out):
    L = lib()
    L.go_intersect_kmer_sets.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.c_void_p, C.c_char_p,
                                         C.c_char_p, C.c_size_t]
    src = FileSet.from_files(files)
    dst = FileSet()
    arr = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
    err = C.create_string_buffer(1024)
    if L.go_intersect_kmer_sets(src.handle, arr, len(names), dst.handle, out.encode(), err, 1024):
        raise OracleError(err.value.decode())
    return dst.files()


def subtract_kmer_set(files, lhs, rhs, out):
    L = lib()
    L.go_subtract_kmer_set.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    src = FileSet.from_files(files)
    dst = FileSet()
    err = C.create_string_buffer(1024)
    if L.go_subtract_kmer_set(src.handle, lhs.encode(), rhs.encode(), dst.handle, out.encode(), err, 1024):
        raise OracleError(err.value.decode())
    return dst.files()


def graph_to_kmer_set(files, graph, out):
    L = lib()
    L.go_graph_to_kmer_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    src = FileSet.from_files(files)
    dst = FileSet()
    err = C.create_string_buffer(1024)
    if L.go_graph_to_kmer_set(src.handle, graph.encode(), dst.handle, out.encode(), err, 1024):
        raise OracleError(err.value.decode())
    return dst.files()


def merge_and_annotate(files, lhs, rhs, out):
    """Returns (files, (lhs count, rhs count, common))."""
    L = lib()
    L.go_merge_and_annotate.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_char_p,
                                        C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    src = FileSet.from_files(files)
    dst = FileSet()
    err = C.create_string_buffer(1024)
    stats = (C.c_uint64 * 3)()
    if L.go_merge_and_annotate(src.handle, lhs.encode(), rhs.encode(), dst.handle, out.encode(), stats, err, 1024):
        raise OracleError(err.value.decode())
    return dst.files(), tuple(stats)


# --------------------------------------------------------------------------------------
# stand-alone structures and assertion replays in the shape of the reference's unit tests
# (goss_oracle.h: go_write_bits_and_select ... go_replay_vba)
# --------------------------------------------------------------------------------------

def _u64_array(vals):
    import numpy as np
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.uint64))
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


def _key_array(positions):
    n = len(positions)
    return (Key * max(1, n))(*[key(v) for v in positions])


def write_bits_and_select(ones, nbits, invert, vname="v", xname="x"):
    """testDenseArray.cc's set-up: WordyBitVector over nbits positions + DenseSelect of one sense."""
    L = lib()
    L.go_write_bits_and_select.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64, C.c_int]
    fs = FileSet()
    a, p = _u64_array(ones)
    if L.go_write_bits_and_select(fs.handle, vname.encode(), xname.encode(), p, len(ones), nbits, 1 if invert else 0):
        raise OracleError("go_write_bits_and_select")
    return fs.files()


def write_bits_sparse(ones, vname="x"):
    L = lib()
    L.go_write_bits_sparse.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t]
    fs = FileSet()
    a, p = _u64_array(ones)
    L.go_write_bits_sparse(fs.handle, vname.encode(), p, len(ones))
    return fs.files()


class BitsReader:
    """WordyBitVector read side (get / select1 / select0 / popcountRange / words)."""

    def __init__(self, files, vname="x"):
        self._fs = FileSet.from_files(files)
        self._v = vname.encode()
        L = lib()
        L.go_bits_get.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.go_bits_words.restype = C.c_uint64
        L.go_bits_words.argtypes = [C.c_void_p, C.c_char_p]
        L.go_bits_select.restype = C.c_uint64
        L.go_bits_select.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_uint64, C.c_uint64]
        L.go_bits_popcount_range.restype = C.c_uint64
        L.go_bits_popcount_range.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64]

    def get(self, pos):
        return bool(lib().go_bits_get(self._fs.handle, self._v, pos))

    def words(self):
        return lib().go_bits_words(self._fs.handle, self._v)

    def select1(self, frm, count):
        return lib().go_bits_select(self._fs.handle, self._v, 0, frm, count)

    def select0(self, frm, count):
        return lib().go_bits_select(self._fs.handle, self._v, 1, frm, count)

    def popcount_range(self, b, e):
        return lib().go_bits_popcount_range(self._fs.handle, self._v, b, e)


def dense_select(files, i, invert, vname="v", xname="x"):
    L = lib()
    L.go_dense_select.restype = C.c_uint64
    L.go_dense_select.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_uint64]
    fs = FileSet.from_files(files)
    return L.go_dense_select(fs.handle, vname.encode(), xname.encode(), 1 if invert else 0, i)


def replay_dense_select(files, ones, nbits, invert, vname="v", xname="x"):
    L = lib()
    L.go_replay_dense_select.restype = C.c_uint64
    L.go_replay_dense_select.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64]
    fs = FileSet.from_files(files)
    a, p = _u64_array(ones)
    return L.go_replay_dense_select(fs.handle, vname.encode(), xname.encode(), 1 if invert else 0, p, len(ones), nbits)


def replay_sparse(files, base, positions, universe=0):
    L = lib()
    L.go_replay_sparse.restype = C.c_uint64
    L.go_replay_sparse.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Key), C.c_size_t, C.c_uint64]
    fs = FileSet.from_files(files)
    return L.go_replay_sparse(fs.handle, base.encode(), _key_array(positions), len(positions), universe)


def replay_sparse_highbits(files, base, ones, nbits):
    L = lib()
    L.go_replay_sparse_highbits.restype = C.c_uint64
    L.go_replay_sparse_highbits.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64]
    fs = FileSet.from_files(files)
    a, p = _u64_array(ones)
    return L.go_replay_sparse_highbits(fs.handle, base.encode(), p, len(ones), nbits)


def write_vba(values, num_items, base="x"):
    import numpy as np
    L = lib()
    L.go_write_vba.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.c_size_t, C.c_uint64]
    fs = FileSet()
    a = np.ascontiguousarray(np.asarray(values, dtype=np.uint32))
    if L.go_write_vba(fs.handle, base.encode(), a.ctypes.data_as(C.POINTER(C.c_uint32)), len(values), num_items):
        raise OracleError("go_write_vba")
    return fs.files()


def replay_vba(files, base, values):
    import numpy as np
    L = lib()
    L.go_replay_vba.restype = C.c_uint64
    L.go_replay_vba.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.c_size_t]
    fs = FileSet.from_files(files)
    a = np.ascontiguousarray(np.asarray(values, dtype=np.uint32))
    return L.go_replay_vba(fs.handle, base.encode(), a.ctypes.data_as(C.POINTER(C.c_uint32)), len(values))


def build_kmer_set_mt(reads, K, threads, out="ks"):
    """go_build_kmer_set_mt: the line-kind read text counted with `threads` worker threads."""
    L = lib()
    L.go_build_kmer_set_mt.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_char_p, C.c_size_t, C.c_uint,
                                       C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    fs = FileSet()
    err = C.create_string_buffer(512)
    nwin = C.c_uint64(0)
    if isinstance(reads, str):
        reads = reads.encode()
    if L.go_build_kmer_set_mt(fs.handle, out.encode(), K, reads, len(reads), threads, C.byref(nwin), err, 512):
        raise OracleError(err.value.decode())
    return fs.files(), nwin.value
