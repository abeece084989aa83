"""ctypes binding of libgossgpu.so (the C ABI declared in include/goss_gpu.h).

This module is plumbing only: it loads the in-tree shared library, declares the entry
points and wraps them in a small context class.  There is no fallback of any kind: if the
library is missing, or no gfx950 device is usable, the calls raise.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOSS_GPU_LIB") or os.path.join(_PKG, "libgossgpu.so")

MODE_KMER_SET = 0
MODE_GRAPH = 1

# every symbol include/goss_gpu.h declares
SYMBOLS = [
    "goss_gpu_strerror", "goss_gpu_last_error", "goss_gpu_abi_version", "goss_gpu_create",
    "goss_gpu_destroy", "goss_gpu_push_bases_host", "goss_gpu_push_bases_device",
    "goss_gpu_finish", "goss_gpu_result", "goss_gpu_result_copy", "goss_gpu_emit",
    "goss_gpu_file_count", "goss_gpu_file_info", "goss_gpu_file_read",
    "goss_gpu_emit_sparse_array", "goss_gpu_timing_get", "goss_gpu_timing_reset",
    "goss_gpu_synth_reads", "goss_synth_reads_host", "goss_gpu_reset", "goss_gpu_push_run_device", "goss_gpu_set_path", "goss_gpu_host_alloc", "goss_gpu_host_free", "goss_gpu_host_register", "goss_gpu_host_unregister", "goss_gpu_push_run_sparse", "goss_gpu_push_run_host", "goss_gpu_emit_estimate",
    "goss_gpu_select_counts", "goss_gpu_select_normal", "goss_gpu_emit_count_bits", "goss_gpu_emit_dump", "goss_gpu_lint", "goss_gpu_stat", "goss_gpu_check_index",
    "goss_gpu_set_budget_limit", "goss_gpu_emit_dump_range", "goss_gpu_prepare", "goss_gpu_emit_part", "goss_gpu_emit_assemble", "goss_gpu_emit_part_ranges", "goss_gpu_emit_last_high",
    "goss_gpu_file_device", "goss_gpu_big_counts", "goss_gpu_push_run_graph",
    "goss_gpu_group_exchange", "goss_gpu_group_emit",
    "goss_gpu_set_deferred", "goss_gpu_stage_room", "goss_gpu_group_route_exchange",
    "goss_gpu_push_keys_host", "goss_gpu_push_keys_device", "goss_gpu_push_packed_device", "goss_gpu_pack_bases_device", "goss_gpu_expect_bases",
    "goss_gpu_route_records_device", "goss_gpu_push_records_device",
    "goss_gpu_push_bases_host_async", "goss_gpu_push_packed_host", "goss_gpu_push_packed_host_async", "goss_gpu_flush",
]

RECORD_BYTES = 12          # one-word keys (2 * len <= 62); two-word keys: 20 (record_bytes)


def record_bytes(k, mode=0):
    """bytes of a super-k-mer record of a (k, mode) context: SkRec (12) for one-word keys, SkRec2 (20) for two-word keys"""
    length = k + (1 if mode == 1 else 0)
    return 12 if 2 * length <= 62 else 20


RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p)


def pack_bases(text):
    """bytes of bases -> (codes u32 per 16 positions, nonbase u16 per 16 positions): the packed form of
    goss_gpu_push_packed_host (numpy restatement of the host parser's packer, for the tests)."""
    import numpy as np
    a = np.frombuffer(text if isinstance(text, (bytes, bytearray)) else text.encode(), dtype=np.uint8)
    n = a.size
    g = (n + 15) // 16
    pad = np.full(g * 16, 10, dtype=np.uint8)
    pad[:n] = a
    low = pad | 0x20
    code = np.zeros(g * 16, dtype=np.uint32)
    code[low == ord("c")] = 1
    code[low == ord("g")] = 2
    code[low == ord("t")] = 3
    bad = ~((low == ord("a")) | (low == ord("c")) | (low == ord("g")) | (low == ord("t")))
    shifts = (2 * np.arange(16, dtype=np.uint32))[None, :]
    codes = np.bitwise_or.reduce(code.reshape(g, 16) << shifts, axis=1).astype(np.uint32)
    bshift = np.arange(16, dtype=np.uint32)[None, :]
    nonbase = np.bitwise_or.reduce(bad.reshape(g, 16).astype(np.uint32) << bshift, axis=1).astype(np.uint16)
    return np.ascontiguousarray(codes), np.ascontiguousarray(nonbase)


class GossGpuError(RuntimeError):
    def __init__(self, status, text, detail=""):
        super().__init__("%s (status %d)%s" % (text, status, (": " + detail) if detail else ""))
        self.status = status


class Counts(C.Structure):
    _fields_ = [("windows", C.c_uint64), ("keys", C.c_uint64), ("distinct", C.c_uint64),
                ("key_words", C.c_uint32), ("reserved", C.c_uint32)]


T_EXTRACT, T_HIST, T_SCAN, T_SCATTER, T_REDUCE, T_EMIT, T_ORDER, T_CLASSES = 0, 1, 2, 3, 4, 5, 6, 8
T_NAMES = ["extract", "hist", "scan", "scatter", "reduce", "emit", "order"]


class Timing(C.Structure):
    _fields_ = [("ms", C.c_float * T_CLASSES), ("launches", C.c_uint32 * T_CLASSES),
                ("units", C.c_uint64 * T_CLASSES)]

    def as_dict(self):
        return {n: {"ms": self.ms[i], "launches": self.launches[i], "units": self.units[i]}
                for i, n in enumerate(T_NAMES)}


_lib = None


def load():
    """Load libgossgpu.so; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GossGpuError(-2, "libgossgpu.so is not built", LIB_PATH)
    try:
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7; importing it
        # first makes libgossgpu.so bind to that same runtime (same SONAME) instead of
        # bringing /opt/rocm's copy in beside it, which leaves torch without a GPU.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.goss_gpu_strerror.restype = C.c_char_p
    L.goss_gpu_strerror.argtypes = [C.c_int]
    L.goss_gpu_last_error.restype = C.c_char_p
    L.goss_gpu_last_error.argtypes = [C.c_void_p]
    L.goss_gpu_abi_version.restype = C.c_uint32
    L.goss_gpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_int, C.c_uint64, C.c_void_p]
    L.goss_gpu_destroy.argtypes = [C.c_void_p]
    L.goss_gpu_destroy.restype = None
    L.goss_gpu_push_bases_host.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.goss_gpu_push_bases_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.goss_gpu_finish.argtypes = [C.c_void_p, C.POINTER(Counts)]
    L.goss_gpu_result.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.goss_gpu_result_copy.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    L.goss_gpu_emit.argtypes = [C.c_void_p]
    L.goss_gpu_file_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.goss_gpu_file_info.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64)]
    L.goss_gpu_file_read.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64]
    L.goss_gpu_emit_sparse_array.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                                             C.c_uint64, C.c_uint64, C.c_uint64]
    L.goss_gpu_timing_get.argtypes = [C.c_void_p, C.POINTER(Timing)]
    L.goss_gpu_timing_reset.argtypes = [C.c_void_p]
    L.goss_gpu_reset.argtypes = [C.c_void_p]
    L.goss_gpu_set_path.argtypes = [C.c_void_p, C.c_int]
    L.goss_gpu_push_run_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.goss_gpu_synth_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64]
    L.goss_synth_reads_host.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64]
    _lib = L
    return L


def synth_reads_host(nreads, read_len, genome_len, seed=1, first_read=0):
    """The synthetic read generator on the host: bytes of nreads*(read_len+1)."""
    L = load()
    buf = C.create_string_buffer(nreads * (read_len + 1))
    rc = L.goss_synth_reads_host(buf, nreads, read_len, genome_len, seed, first_read)
    if rc:
        raise GossGpuError(rc, L.goss_gpu_strerror(rc).decode())
    return buf.raw


def _torch_ready():
    """The library runs on its own HIP stream.  Device buffers handed to it by pointer usually come
    from torch, whose kernels are asynchronous on torch's stream: wait for them (C callers of the
    ABI order their own streams; this wrapper does it for the Python tests and bench.py)."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()


def group_exchange(contexts, sample_per_context=0):
    """goss_gpu_group_exchange on several finished Contexts of one process: context j ends up holding range j
    of the union.  Returns the range sizes."""
    L = load()
    n = len(contexts)
    arr = (C.c_void_p * n)(*[c._h for c in contexts])
    sizes = (C.c_uint64 * n)()
    L.goss_gpu_group_exchange.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    _torch_ready()
    contexts[0]._check(L.goss_gpu_group_exchange(arr, n, sample_per_context, sizes))
    return [int(x) for x in sizes]


class GroupXStats(C.Structure):
    _fields_ = [("transport", C.c_uint32), ("rounds", C.c_uint32), ("records", C.c_uint64), ("windows", C.c_uint64),
                ("record_bytes", C.c_uint64), ("route_ms", C.c_double), ("wire_ms", C.c_double), ("count_ms", C.c_double),
                ("count_wait_ms", C.c_double)]


def group_route_exchange(contexts, transport=0):
    """goss_gpu_group_route_exchange: what the (deferred) contexts have staged is cut into super-k-mer records routed by
    minimizer, part p of every context moved to context p and counted there.  Returns the call's statistics as a dict."""
    L = load()
    n = len(contexts)
    arr = (C.c_void_p * n)(*[c._h for c in contexts])
    st = GroupXStats()
    L.goss_gpu_group_route_exchange.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_int, C.POINTER(GroupXStats)]
    _torch_ready()
    contexts[0]._check(L.goss_gpu_group_route_exchange(arr, n, transport, C.byref(st)))
    return {f: getattr(st, f) for f, _ in GroupXStats._fields_}


def group_emit(contexts, estimate=0):
    """goss_gpu_group_emit: every context's slices, the index files on contexts[0]"""
    L = load()
    n = len(contexts)
    arr = (C.c_void_p * n)(*[c._h for c in contexts])
    L.goss_gpu_group_emit.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint64]
    contexts[0]._check(L.goss_gpu_group_emit(arr, n, estimate))


class Context:
    """One counting context on one GPU (include/goss_gpu.h)."""

    def __init__(self, k, mode=MODE_KMER_SET, device=0, hbm_budget=0, stream=None):
        self._L = load()
        self._h = C.c_void_p()
        rc = self._L.goss_gpu_create(C.byref(self._h), device, k, mode, hbm_budget, stream)
        if rc:
            self._h = None
            raise GossGpuError(rc, self._L.goss_gpu_strerror(rc).decode())
        self.k = k
        self.mode = mode
        self.key_words = 1 if 2 * (k + (1 if mode == MODE_GRAPH else 0)) <= 62 else 2

    def close(self):
        if getattr(self, "_h", None):
            self._L.goss_gpu_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc:
            raise GossGpuError(rc, self._L.goss_gpu_strerror(rc).decode(),
                               self._L.goss_gpu_last_error(self._h).decode())

    def push_host(self, data):
        if isinstance(data, str):
            data = data.encode()
        self._check(self._L.goss_gpu_push_bases_host(self._h, data, len(data)))

    def push_host_async(self, data, on_release=None):
        """goss_gpu_push_bases_host_async: the bytes object stays referenced until the library hands it back."""
        if isinstance(data, str):
            data = data.encode()
        if not hasattr(self, "_lent"):
            self._lent, self._lent_id = {}, 0
            self._release_cb = RELEASE_FN(self._released)
        self._lent_id += 1
        self._lent[self._lent_id] = (data, on_release)
        self._L.goss_gpu_push_bases_host_async.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, RELEASE_FN, C.c_void_p]
        self._check(self._L.goss_gpu_push_bases_host_async(self._h, data, len(data), self._release_cb, C.c_void_p(self._lent_id)))

    def _released(self, user):
        data, cb = self._lent.pop(int(user or 0))
        if cb:
            cb()

    def push_packed_host(self, text, async_=False):
        """Pack a byte string of bases on the host (2 bits per base + 1 flag bit) and push it
        (goss_gpu_push_packed_host / _async).  Returns (codes, nonbase) numpy arrays."""
        import numpy as np
        codes, bad = pack_bases(text)
        n = len(text)
        if async_:
            if not hasattr(self, "_lent"):
                self._lent, self._lent_id = {}, 0
                self._release_cb = RELEASE_FN(self._released)
            self._lent_id += 1
            self._lent[self._lent_id] = ((codes, bad), None)
            self._L.goss_gpu_push_packed_host_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, RELEASE_FN, C.c_void_p]
            self._check(self._L.goss_gpu_push_packed_host_async(self._h, C.c_void_p(codes.ctypes.data), C.c_void_p(bad.ctypes.data), n,
                                                                self._release_cb, C.c_void_p(self._lent_id)))
        else:
            self._L.goss_gpu_push_packed_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
            self._check(self._L.goss_gpu_push_packed_host(self._h, C.c_void_p(codes.ctypes.data), C.c_void_p(bad.ctypes.data), n))
        return codes, bad

    def flush(self):
        """goss_gpu_flush: wait for the queued copies; every lent buffer comes back."""
        self._L.goss_gpu_flush.argtypes = [C.c_void_p]
        self._check(self._L.goss_gpu_flush(self._h))

    def push_device(self, ptr, nbytes):
        _torch_ready()
        self._check(self._L.goss_gpu_push_bases_device(self._h, C.c_void_p(ptr), nbytes))

    def expect_bases(self, total_bases):
        """goss_gpu_expect_bases: a hint -- how many bases will be pushed in all before finish (0: unknown)."""
        self._L.goss_gpu_expect_bases.argtypes = [C.c_void_p, C.c_uint64]
        self._check(self._L.goss_gpu_expect_bases(self._h, total_bases))

    def push_packed_device(self, codes_ptr, nonbase_ptr, nbases):
        """goss_gpu_push_packed_device: packed bases resident in HBM (u32 of codes + u16 of flags per 16 positions)."""
        _torch_ready()
        self._L.goss_gpu_push_packed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        self._check(self._L.goss_gpu_push_packed_device(self._h, C.c_void_p(codes_ptr), C.c_void_p(nonbase_ptr), nbases))

    def pack_bases_device(self, bases_ptr, nbytes, codes_ptr, nonbase_ptr):
        """goss_gpu_pack_bases_device: bytes in HBM -> the packed form in the caller's arrays (ceil(nbytes / 16) each)."""
        _torch_ready()
        self._L.goss_gpu_pack_bases_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        self._check(self._L.goss_gpu_pack_bases_device(self._h, C.c_void_p(bases_ptr), nbytes, C.c_void_p(codes_ptr), C.c_void_p(nonbase_ptr)))

    def push_run(self, keys_ptr, counts_ptr, m):
        _torch_ready()
        self._check(self._L.goss_gpu_push_run_device(self._h, C.c_void_p(keys_ptr), C.c_void_p(counts_ptr), m))

    def push_run_host(self, keys_ptr, counts_ptr, m):
        """A counted run in host memory (goss_gpu_push_run_host): m keys of key_words u64, m u32 counts."""
        self._L.goss_gpu_push_run_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        self._check(self._L.goss_gpu_push_run_host(self._h, C.c_void_p(keys_ptr), C.c_void_p(counts_ptr), m))

    def push_keys_host(self, keys):
        """Raw k-mers (unsorted, un-normalised) in host memory: a numpy uint64 array of n * key_words values
        (goss_gpu_push_keys_host)."""
        import numpy as np
        a = np.ascontiguousarray(keys, dtype=np.uint64)
        n = a.size // self.key_words
        self._L.goss_gpu_push_keys_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        self._check(self._L.goss_gpu_push_keys_host(self._h, C.c_void_p(a.ctypes.data), n))

    def push_keys_device(self, ptr, n):
        """Raw k-mers resident in HBM (goss_gpu_push_keys_device)."""
        _torch_ready()
        self._L.goss_gpu_push_keys_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        self._check(self._L.goss_gpu_push_keys_device(self._h, C.c_void_p(ptr), n))

    def route_records(self, bases_ptr, nbytes, nparts, records_ptr, part_first, part_cap, ready=False):
        """goss_gpu_route_records_device: the windows of a device-resident base string as super-k-mer records in
        nparts buffers.  Returns (record slots filled per part -- pads, records of no window, included --, windows per
        part, ok); ok False = some part_cap was too small and `records` holds what every part needs.  ready: the
        caller has waited for whatever wrote the bases (no device-wide synchronisation here)."""
        if not ready:
            _torch_ready()
        U = C.c_uint64 * nparts
        first, cap, recs, wins = U(*part_first), U(*part_cap), U(), U()
        self._L.goss_gpu_route_records_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                          C.POINTER(C.c_uint64)]
        rc = self._L.goss_gpu_route_records_device(self._h, C.c_void_p(bases_ptr), nbytes, nparts, C.c_void_p(records_ptr),
                                                   first, cap, recs, wins)
        if rc not in (0, -9):
            self._check(rc)
        return [int(x) for x in recs], [int(x) for x in wins], rc == 0

    def push_records(self, records_ptr, nrecords, nwindows=0, ready=False):
        """goss_gpu_push_records_device: count the windows of super-k-mer records resident in HBM.  ready: the caller
        has waited for whatever wrote the records (no device-wide synchronisation here: other streams go on)."""
        if not ready:
            _torch_ready()
        self._L.goss_gpu_push_records_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        self._check(self._L.goss_gpu_push_records_device(self._h, C.c_void_p(records_ptr), nrecords, nwindows))

    def prepare(self):
        """Start mapping the HBM arena in the background (goss_gpu_prepare)."""
        self._L.goss_gpu_prepare.argtypes = [C.c_void_p]
        self._check(self._L.goss_gpu_prepare(self._h))

    def push_run_graph(self, files, key_bits):
        """A graph given as {suffix: bytes} of its files ("-edges.*", "-counts.*") as one run, edges and
        multiplicities decoded on the device (goss_gpu_push_run_graph)."""
        import struct

        class SparseRun(C.Structure):
            _fields_ = [("D", C.c_uint64), ("count", C.c_uint64), ("high_bits", C.c_char_p), ("high_words", C.c_uint64),
                        ("ncols", C.c_uint32), ("weight", C.c_uint32), ("col", C.c_char_p * 4), ("col_bytes", C.c_uint32 * 4),
                        ("col_shift", C.c_uint32 * 4), ("counts", C.c_void_p)]

        class Vba(C.Structure):
            _fields_ = [("ord0", C.c_char_p), ("ord0_bytes", C.c_uint64), ("ord1p", SparseRun), ("ord1", C.c_char_p),
                        ("ord1_bytes", C.c_uint64), ("ord2p", SparseRun), ("ord2", C.c_char_p), ("ord2_bytes", C.c_uint64)]

        def layout(bits, prefix, shift, out):
            split = {24: (8, 16), 40: (8, 32), 48: (16, 32), 56: (8, 48), 72: (8, 64), 80: (16, 64), 88: (8, 80),
                     96: (32, 64), 104: (8, 96), 112: (16, 96), 120: (24, 96), 128: (64, 64)}
            if bits in (8, 16, 32, 64):
                out.append((prefix, bits // 8, shift))
                return
            ub, lb = split[bits]
            layout(ub, prefix + ".upr", shift + lb, out)
            layout(lb, prefix + ".lwr", shift, out)

        keep = []

        def run(base):
            h = struct.unpack("<8Q", files[base + ".header"][:64])
            r = SparseRun()
            r.D, r.count = h[1], h[7]
            hb = files[base + ".high-bits"]
            keep.append(hb)
            r.high_bits, r.high_words = hb, len(hb) // 8
            cols = []
            layout(h[2], "", 0, cols)
            r.ncols = len(cols)
            for i, (suffix, nbytes, shift) in enumerate(cols):
                data = files[base + ".low-bits" + suffix] or b"\0"
                keep.append(data)
                r.col[i], r.col_bytes[i], r.col_shift[i] = data, nbytes, shift
            return r

        edges = run("-edges")
        v = Vba()
        v.ord0, v.ord0_bytes = files["-counts.ord0"] or b"\0", len(files["-counts.ord0"])
        v.ord1p = run("-counts.ord1p")
        v.ord1, v.ord1_bytes = files["-counts.ord1"] or b"\0", len(files["-counts.ord1"])
        v.ord2p = run("-counts.ord2p")
        v.ord2, v.ord2_bytes = files["-counts.ord2"] or b"\0", len(files["-counts.ord2"])
        self._L.goss_gpu_push_run_graph.argtypes = [C.c_void_p, C.POINTER(SparseRun), C.POINTER(Vba)]
        self._check(self._L.goss_gpu_push_run_graph(self._h, C.byref(edges), C.byref(v)))

    def set_path(self, path):
        """0: segment hash path with LSD fallback (default); 1: LSD radix sort only."""
        self._check(self._L.goss_gpu_set_path(self._h, path))

    def reset(self):
        self._check(self._L.goss_gpu_reset(self._h))

    def set_budget_limit(self, max_bytes):
        """Let the arena grow up to max_bytes when a chunk or a merge needs it (goss_gpu_set_budget_limit)."""
        self._L.goss_gpu_set_budget_limit.argtypes = [C.c_void_p, C.c_uint64]
        self._check(self._L.goss_gpu_set_budget_limit(self._h, max_bytes))

    def set_deferred(self, on=True):
        """goss_gpu_set_deferred: host pushes only stage; a full staging buffer is reported (GOSS_ERR_BUFFER)."""
        self._L.goss_gpu_set_deferred.argtypes = [C.c_void_p, C.c_int]
        self._check(self._L.goss_gpu_set_deferred(self._h, 1 if on else 0))

    def stage_room(self):
        """goss_gpu_stage_room -> (bytes the staging buffer still takes, its capacity)"""
        a, b = C.c_uint64(), C.c_uint64()
        self._L.goss_gpu_stage_room.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self._check(self._L.goss_gpu_stage_room(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stat(self, name):
        """A diagnostic counter of the context (goss_gpu_stat)."""
        v = C.c_uint64()
        self._L.goss_gpu_stat.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        self._check(self._L.goss_gpu_stat(self._h, name.encode(), C.byref(v)))
        return v.value

    def finish(self):
        c = Counts()
        self._check(self._L.goss_gpu_finish(self._h, C.byref(c)))
        self.counts = c
        return c

    def result_ptrs(self):
        k, v, m = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self._L.goss_gpu_result(self._h, C.byref(k), C.byref(v), C.byref(m)))
        return k.value, v.value, m.value

    def result(self):
        """(keys as python ints, counts) copied to the host."""
        import numpy as np
        m = self.counts.distinct
        w = self.counts.key_words
        keys = np.zeros(max(1, m * w), dtype=np.uint64)
        cnts = np.zeros(max(1, m), dtype=np.uint32)
        self._check(self._L.goss_gpu_result_copy(self._h, 0, m, keys.ctypes.data_as(C.c_void_p), cnts.ctypes.data_as(C.c_void_p)))
        keys = keys[: m * w]
        cnts = cnts[:m]
        if w == 1:
            ks = keys.tolist()
        else:
            ks = [l | (h << 64) for l, h in zip(keys[0::2].tolist(), keys[1::2].tolist())]
        return ks, cnts

    def big_counts(self):
        """{key: exact count} of the keys that occurred 2^32 - 1 times or more (goss_gpu_big_counts)."""
        self._L.goss_gpu_big_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
        n = C.c_uint32()
        keys = (C.c_uint64 * 512)()
        counts = (C.c_uint64 * 256)()
        self._check(self._L.goss_gpu_big_counts(self._h, keys, counts, 256, C.byref(n)))
        return {int(keys[2 * i]) | (int(keys[2 * i + 1]) << 64): int(counts[i]) for i in range(min(n.value, 256))}

    def select_counts(self, lo, hi):
        """Between finish and emit: keep the result items whose count lies in [lo, hi]
        (goss_gpu_select_counts: the set algebra of intersect-kmer-sets / subtract-kmer-set)."""
        self._L.goss_gpu_select_counts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        self._check(self._L.goss_gpu_select_counts(self._h, lo, hi))

    def select_normal(self):
        """Between finish and emit: keep the items that are their own canonical form (goss_gpu_select_normal)."""
        self._L.goss_gpu_select_normal.argtypes = [C.c_void_p]
        self._check(self._L.goss_gpu_select_normal(self._h))

    def emit_device(self):
        """Build the on-disk arrays in HBM without copying them to the host."""
        self._check(self._L.goss_gpu_emit(self._h))

    def emit(self):
        self.emit_device()
        return self.files()

    def emit_part(self, first_index, total, estimate=0, prev_last_high=None):
        """Distributed emission, this range's part (goss_gpu_emit_part; with prev_last_high -- the high part of the last
        key of the ranges below, emit_last_high of their owners -- goss_gpu_emit_part_ranges: "-d0" blocks too)."""
        if prev_last_high is None:
            self._L.goss_gpu_emit_part.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
            self._check(self._L.goss_gpu_emit_part(self._h, first_index, total, estimate))
        else:
            self._L.goss_gpu_emit_part_ranges.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
            self._check(self._L.goss_gpu_emit_part_ranges(self._h, first_index, total, estimate, prev_last_high))

    def emit_last_high(self, total, estimate=0):
        """goss_gpu_emit_last_high -> (high part of this range's last key, range is not empty)"""
        h, ne = C.c_uint64(), C.c_int()
        self._L.goss_gpu_emit_last_high.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        self._check(self._L.goss_gpu_emit_last_high(self._h, total, estimate, C.byref(h), C.byref(ne)))
        return h.value, bool(ne.value)

    def emit_assemble(self, spans_ptr, span_bytes, total, estimate=0, big=b"", hist=b""):
        """Distributed emission, the files that need all ranges (goss_gpu_emit_assemble): spans_ptr =
        device address of the ranges' ".part.span" files back to back (span_bytes in all), big / hist = the
        concatenated records (bytes)."""
        self._L.goss_gpu_emit_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                                   C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        _torch_ready()
        self._check(self._L.goss_gpu_emit_assemble(self._h, C.c_void_p(spans_ptr), span_bytes, total, estimate,
                                                   big, len(big) // 16, hist, len(hist) // 16))

    def file_list(self):
        """[(suffix, size, device address or None)] of the emitted files"""
        self._L.goss_gpu_file_device.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        n = C.c_uint32()
        self._check(self._L.goss_gpu_file_count(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.create_string_buffer(256)
            size = C.c_uint64()
            self._check(self._L.goss_gpu_file_info(self._h, i, name, 256, C.byref(size)))
            p = C.c_void_p()
            self._check(self._L.goss_gpu_file_device(self._h, i, C.byref(p)))
            out.append((name.value.decode(), size.value, p.value))
        return out

    def read_file(self, suffix):
        """bytes of one emitted file"""
        for i, (name, size, _) in enumerate(self.file_list()):
            if name == suffix:
                buf = C.create_string_buffer(max(1, size))
                self._check(self._L.goss_gpu_file_read(self._h, i, 0, buf, size))
                return buf.raw[:size]
        raise KeyError(suffix)

    def emit_sparse_array(self, dev_ptr, key_words, n, N, M, N_end=None):
        if N_end is None:
            N_end = N
        mask = (1 << 64) - 1
        _torch_ready()
        self._check(self._L.goss_gpu_emit_sparse_array(self._h, C.c_void_p(dev_ptr), key_words, n, N & mask, N >> 64, M,
                                                       N_end & mask, N_end >> 64))
        return self.files()

    def lint(self, asymmetric=False):
        """goss_gpu_lint on the finished graph: dict of the problem counts of lint-graph's pass 1."""
        class LintReport(C.Structure):
            _fields_ = [("missing_rc", C.c_uint64), ("count_mismatch", C.c_uint64), ("zero_count", C.c_uint64),
                        ("order_violation", C.c_uint64), ("nexamples", C.c_uint32), ("pad", C.c_uint32),
                        ("ex_index", C.c_uint64 * 32), ("ex_other", C.c_uint64 * 32), ("ex_kind", C.c_uint32 * 32)]
        rep = LintReport()
        self._L.goss_gpu_lint.argtypes = [C.c_void_p, C.c_int, C.POINTER(LintReport)]
        self._check(self._L.goss_gpu_lint(self._h, 1 if asymmetric else 0, C.byref(rep)))
        return {"missing_rc": rep.missing_rc, "count_mismatch": rep.count_mismatch, "zero_count": rep.zero_count,
                "order_violation": rep.order_violation}

    def check_index(self, files, base=""):
        """goss_gpu_check_index on a SparseArray given as {suffix: bytes} (files[base + ".header"]
        ...): the context must hold the array's decoded elements (push_run + finish).  Returns a
        dict of the mismatch counts."""
        import struct

        def layout(bits, prefix, shift, out):
            # IntegerArray::builder column files (IntegerArray.cc:259-357)
            split = {24: (8, 16), 40: (8, 32), 48: (16, 32), 56: (8, 48), 72: (8, 64), 80: (16, 64), 88: (8, 80),
                     96: (32, 64), 104: (8, 96), 112: (16, 96), 120: (24, 96), 128: (64, 64)}
            if bits in (8, 16, 32, 64):
                out.append((prefix, bits // 8, shift))
                return
            ub, lb = split[bits]
            layout(ub, prefix + ".upr", shift + lb, out)
            layout(lb, prefix + ".lwr", shift, out)

        class SparseFiles(C.Structure):
            _fields_ = [("D", C.c_uint64), ("count", C.c_uint64), ("size_lo", C.c_uint64), ("size_hi", C.c_uint64),
                        ("high_bits", C.c_char_p), ("high_words", C.c_uint64),
                        ("d0", C.c_char_p), ("d0_bytes", C.c_uint64), ("d1", C.c_char_p), ("d1_bytes", C.c_uint64),
                        ("ncols", C.c_uint32), ("pad", C.c_uint32),
                        ("col", C.c_char_p * 4), ("col_bytes", C.c_uint32 * 4), ("col_shift", C.c_uint32 * 4)]

        class IndexReport(C.Structure):
            _fields_ = [("select_mismatch", C.c_uint64), ("rank_mismatch", C.c_uint64), ("access_miss", C.c_uint64),
                        ("failures", C.c_uint64), ("nexamples", C.c_uint32), ("pad", C.c_uint32),
                        ("ex_index", C.c_uint64 * 16), ("ex_kind", C.c_uint32 * 16)]

        h = struct.unpack("<8Q", files[base + ".header"][:64])
        f = SparseFiles()
        f.D, qd, f.size_lo, f.size_hi, f.count = h[1], h[2], h[5], h[6], h[7]
        hb = files[base + ".high-bits"]
        f.high_bits, f.high_words = hb, len(hb) // 8
        f.d0, f.d0_bytes = files[base + "-d0"], len(files[base + "-d0"])
        f.d1, f.d1_bytes = files[base + "-d1"], len(files[base + "-d1"])
        cols = []
        layout(qd, "", 0, cols)
        f.ncols = len(cols)
        for i, (suffix, nbytes, shift) in enumerate(cols):
            f.col[i] = files[base + ".low-bits" + suffix] or b"\0"
            f.col_bytes[i] = nbytes
            f.col_shift[i] = shift
        rep = IndexReport()
        self._L.goss_gpu_check_index.argtypes = [C.c_void_p, C.POINTER(SparseFiles), C.POINTER(IndexReport)]
        self._check(self._L.goss_gpu_check_index(self._h, C.byref(f), C.byref(rep)))
        return {"select": rep.select_mismatch, "rank": rep.rank_mismatch, "access": rep.access_miss, "failures": rep.failures,
                "examples": [(rep.ex_index[i], rep.ex_kind[i]) for i in range(rep.nexamples)]}

    def files(self):
        n = C.c_uint32()
        self._check(self._L.goss_gpu_file_count(self._h, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = C.create_string_buffer(256)
            size = C.c_uint64()
            self._check(self._L.goss_gpu_file_info(self._h, i, name, 256, C.byref(size)))
            buf = C.create_string_buffer(max(1, size.value))
            self._check(self._L.goss_gpu_file_read(self._h, i, 0, buf, size.value))
            out[name.value.decode()] = buf.raw[: size.value]
        return out

    def timing(self, reset=False):
        t = Timing()
        self._check(self._L.goss_gpu_timing_get(self._h, C.byref(t)))
        if reset:
            self._check(self._L.goss_gpu_timing_reset(self._h))
        return t

    def synth_reads(self, dev_ptr, nreads, read_len, genome_len, seed=1, first_read=0):
        self._check(self._L.goss_gpu_synth_reads(self._h, C.c_void_p(dev_ptr), nreads, read_len, genome_len, seed, first_read))
