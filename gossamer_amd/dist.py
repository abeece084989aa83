"""Multi-GPU counting: range-partition the sorted key space over the ranks of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Each
rank counts its own share of the reads, then ONE exchange step moves every (key,count) pair
to the rank that owns the key's range -- an all-to-all(v) in which each rank sends 1/P of its
distinct keys to each peer, so all xGMI links carry traffic at once -- and the owner merges
what it received.  An all-gather of the per-range distinct counts gives the global M (which
fixes the Elias-Fano split D) and each range's rank offset.  There is no ring all-reduce of
bulk data anywhere.  The reference has no distributed path; this module
is new design, constrained only by having to produce the reference's single-pass result.

The functions below work on torch tensors and a process group only, so the same code runs
under gloo on CPU (tests, world_size 2) and under RCCL on GPUs (bench.py).  One-word keys
(2*len <= 62 bits) are int64 tensors of shape [m]: their sign bit is never set, so signed
comparisons order them correctly.  Two-word keys (2*len <= 128) are int64 tensors of shape
[m, 2] with columns (lo, hi) -- the library's Key2 layout; both words are unsigned, so they are
compared after flipping the sign bit.
"""
import torch
import torch.distributed as dist


_SIGN = -(1 << 63)


def _as_i64(v):
    """unsigned 64-bit value -> the int64 with the same bits"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def uniform_splitters(key_bits, parts, dtype=torch.int64, device="cpu"):
    """parts-1 interior splitters cutting [0, 2^key_bits) into equal ranges: shape [parts-1] for
    one-word keys (key_bits <= 62), [parts-1, 2] = (lo, hi) for two-word keys.  Hash-chosen
    canonical k-mers of an i.i.d. genome are uniform on the top bits;
    sampled splitters can replace this for skewed data without changing anything else."""
    total = 1 << key_bits
    cuts = [(total * p) // parts for p in range(1, parts)]
    if key_bits <= 62:
        return torch.tensor(cuts, dtype=dtype, device=device)
    rows = [[_as_i64(c), _as_i64(c >> 64)] for c in cuts]
    return torch.tensor(rows, dtype=dtype, device=device).reshape(len(rows), 2)


def _lower_bound2(keys, lo, hi):
    """number of two-word keys (sorted, columns lo / hi, unsigned) below (hi, lo)"""
    khi = keys[:, 1] ^ _SIGN
    probe = torch.tensor([hi ^ _SIGN], dtype=torch.int64, device=keys.device)
    a = int(torch.searchsorted(khi, probe, right=False).item())
    b = int(torch.searchsorted(khi, probe, right=True).item())
    if a == b:
        return a
    klo = (keys[a:b, 0] ^ _SIGN).contiguous()
    probe = torch.tensor([lo ^ _SIGN], dtype=torch.int64, device=keys.device)
    return a + int(torch.searchsorted(klo, probe, right=False).item())


def split_sizes(sorted_keys, splitters):
    """How many of this rank's sorted distinct keys fall in each range."""
    n = int(sorted_keys.shape[0])
    if splitters.shape[0] == 0:
        return [n]
    if sorted_keys.dim() == 1:
        cuts = torch.searchsorted(sorted_keys, splitters, right=False).tolist()
    else:
        cuts = [_lower_bound2(sorted_keys, int(lo), int(hi)) for lo, hi in splitters.tolist()]
    edges = [0] + cuts + [n]
    return [edges[i + 1] - edges[i] for i in range(len(edges) - 1)]


def exchange_runs(keys, counts, splitters, group=None):
    """all-to-all(v): send range p of (keys, counts) to rank p.
    Returns (recv_keys, recv_counts, recv_sizes): the concatenation of one sorted run per
    source rank, and the run lengths."""
    world = dist.get_world_size(group)
    send = split_sizes(keys, splitters)
    assert len(send) == world
    send_t = torch.tensor(send, dtype=torch.int64, device=keys.device)
    recv_t = torch.empty(world, dtype=torch.int64, device=keys.device)
    dist.all_to_all_single(recv_t, send_t, group=group)
    recv = [int(x) for x in recv_t.tolist()]
    rk = torch.empty((sum(recv),) + tuple(keys.shape[1:]), dtype=keys.dtype, device=keys.device)
    rc = torch.empty(sum(recv), dtype=counts.dtype, device=counts.device)
    dist.all_to_all_single(rk, keys, recv, send, group=group)        # splits along dim 0: whole keys
    dist.all_to_all_single(rc, counts, recv, send, group=group)
    return rk, rc, recv


def gather_counts(m_local, device, group=None):
    """all-gather of the per-range distinct counts -> (list M_p, global M, this rank's offset)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.tensor([m_local], dtype=torch.int64, device=device)
    allm = [torch.empty(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allm, mine, group=group)
    ms = [int(x.item()) for x in allm]
    return ms, sum(ms), sum(ms[:rank])


def gather_ranges_to_root(keys, counts, ms, group=None):
    """Concatenate every rank's range on rank 0 (ranges are disjoint and ordered by rank, so
    the concatenation is the globally sorted distinct key set)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    send = [int(keys.shape[0])] + [0] * (world - 1)
    recv = ms if rank == 0 else [0] * world
    rk = torch.empty((sum(recv),) + tuple(keys.shape[1:]), dtype=keys.dtype, device=keys.device)
    rc = torch.empty(sum(recv), dtype=counts.dtype, device=counts.device)
    dist.all_to_all_single(rk, keys, recv, send, group=group)
    dist.all_to_all_single(rc, counts, recv, send, group=group)
    return rk, rc


class DevArray:
    """Zero-copy torch view of device memory owned by libgossgpu.so."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_view(ptr, n, dtype, device):
    typestr = {torch.int64: "<i8", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(DevArray(ptr, n, typestr), device=device)


def key_view(ptr, m, words, device):
    """the library's key array: [m] int64 for one-word keys, [m, 2] (lo, hi) for two-word keys"""
    v = device_view(ptr, m * words, torch.int64, device)
    return v if words == 1 else v.reshape(m, 2)


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def count_range(ctx, bases_ptr, nbytes, key_bits, device, group=None):
    """local count -> exchange -> merge of the received runs: this rank's range of the global
    result stays in the Context.  Returns (this rank's windows, key words)."""
    world = dist.get_world_size(group)
    ctx.reset()
    ctx.push_device(bases_ptr, nbytes)
    c = ctx.finish()
    words = c.key_words
    windows = c.windows
    kp, cp, m = ctx.result_ptrs()
    keys = key_view(kp, m, words, device)
    counts = device_view(cp, m, torch.int32, device)
    splitters = uniform_splitters(key_bits, world, device=device)
    if (splitters.dim() == 2) != (words == 2):
        raise ValueError("key_bits = %d does not match the context's %d-word keys (2*len: len = k, or k+1 for graphs)" % (key_bits, words))
    rk, rc, recv = exchange_runs(keys, counts, splitters, group)
    _sync(device)          # the library runs on its own stream: finish the collectives first
    # merge the received runs: they replace the local result
    ctx.reset()
    off = 0
    for n in recv:
        if n:
            ctx.push_run(rk.data_ptr() + off * 8 * words, rc.data_ptr() + off * 4, n)
        off += n
    ctx.finish()
    del rk, rc             # the runs were copied into the library's arena
    return windows, words


def _range_tensors(ctx, words, device):
    kp, cp, m = ctx.result_ptrs()
    return key_view(kp, m, words, device).clone(), device_view(cp, m, torch.int32, device).clone()


def _assemble_on_root(ctx, keys, counts, ms, M, device, group):
    """ranges -> rank 0 -> one run -> on-disk arrays (in HBM) on rank 0"""
    ak, ac = gather_ranges_to_root(keys, counts, ms, group)
    _sync(device)
    if dist.get_rank(group) == 0:
        ctx.reset()
        if M:
            ctx.push_run(ak.data_ptr(), ac.data_ptr(), M)
        ctx.finish()
        ctx.emit_device()


def count_distributed(ctx, bases_ptr, nbytes, key_bits, device, group=None, emit_on_root=True):
    """The whole multi-GPU job on an already created Context (either mode, one- or two-word keys):
    local count -> exchange -> merge own range -> all-gather M -> (root) assemble + emit.
    Returns dict(windows=<this rank's windows>, M=<global distinct>, m_range=<this range>)."""
    windows, words = count_range(ctx, bases_ptr, nbytes, key_bits, device, group)
    m_range = ctx.counts.distinct
    ms, M, _ = gather_counts(m_range, device, group)
    out = {"windows": windows, "M": M, "m_range": m_range}
    if emit_on_root:
        keys, counts = _range_tensors(ctx, words, device)
        _assemble_on_root(ctx, keys, counts, ms, M, device, group)
    return out


def set_algebra_distributed(ctx, inputs, key_bits, op, device, group=None, emit_on_root=True):
    """BASELINE config C5 across ranks: every input (bases_ptr, nbytes) is this rank's share of the
    reads of one k-mer set.  The sets are counted and range-partitioned one after the other with
    the same splitters, so the set operation needs no further exchange: each rank combines its
    ranges (goss_gpu_select_counts on weighted runs, as the single-GPU commands do), rank 0
    assembles the result.  op = "intersect" (all sets; globally empty ones are skipped, as
    GossCmdIntersectKmerSets.cc:29-79 does) or "subtract" (first minus second,
    GossCmdSubtractKmerSet.cc:47-66).  Returns dict(sizes=<global size of every input>, M=<result>)."""
    if op not in ("intersect", "subtract") or (op == "subtract" and len(inputs) != 2):
        raise ValueError("op must be 'intersect' or 'subtract' (exactly two sets)")
    ranges, sizes, words = [], [], 1
    for ptr, nbytes in inputs:
        _, words = count_range(ctx, ptr, nbytes, key_bits, device, group)
        ranges.append(_range_tensors(ctx, words, device))
        sizes.append(gather_counts(ranges[-1][0].shape[0], device, group)[1])
    ctx.reset()
    if op == "intersect":
        use = [i for i, n in enumerate(sizes) if n]
        weights = {i: 1 for i in use}
        keep = max(len(use), 1)
    else:
        use = [0, 1]
        weights = {0: 1, 1: 2}
        keep = 1
    held = []
    for i in use:
        keys, _ = ranges[i]
        if keys.shape[0]:
            w = torch.full((keys.shape[0],), weights[i], dtype=torch.int32, device=device)
            held.append(w)
            ctx.push_run(keys.data_ptr(), w.data_ptr(), keys.shape[0])
    _sync(device)
    ctx.finish()
    ctx.select_counts(keep, keep)
    kp, cp, m = ctx.result_ptrs()
    ms, M, _ = gather_counts(m, device, group)
    if emit_on_root:
        keys = key_view(kp, m, words, device).clone()
        counts = torch.ones(m, dtype=torch.int32, device=device)
        _assemble_on_root(ctx, keys, counts, ms, M, device, group)
    return {"sizes": sizes, "M": M}
