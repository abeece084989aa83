"""Multi-GPU counting: range-partition the sorted key space over the ranks of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  Each
rank counts its own share of the reads, then ONE exchange step moves every (key,count) pair
to the rank that owns the key's range -- an all-to-all(v) in which each rank sends 1/P of its
distinct keys to each peer, so all xGMI links carry traffic at once -- and the owner merges
what it received.  An all-gather of the per-range distinct counts gives the global M (which
fixes the Elias-Fano split D) and each range's rank offset.  There is no ring all-reduce of
bulk data anywhere.  The reference has no distributed path; this module
is new design, constrained only by having to produce the reference's single-pass result.

Two forms of the first step (count_range, `exchange=`):
  "counted"  every rank counts ITS reads, then the distinct (key,count) pairs are range-partitioned (what is described
             above).  Cheap on the wire, but with P ranks a rank's local distinct set approaches the whole k-mer set.
  "records"  the exchange comes BEFORE counting: every rank cuts its reads into super-k-mer records routed by
             minimizer (goss_gpu_route_records_device: ~2 bytes per window instead of 8), ONE all-to-all(v) moves
             them, and each rank counts only the keys of its minimizer class -- 1/P of the key space at the
             single-GPU cost, whatever P is.  The classes are not key ranges, so the (now final, disjoint) counted
             sets are range-partitioned afterwards by the same splitter exchange, which then moves 1/P as much.
             12-byte records for one-word keys (2*len <= 62), 20-byte records for two-word keys (len <= 63: the
             minimizer of a long window is taken over its central 31 / 30 bases).

Per exchange of counted runs there are three collectives and one host synchronisation: an all-gather of the
P x P matrix of run lengths (every rank then knows what it receives), and an all-to-all(v) each for the keys and
the counts, sent straight from the library's result arrays (sorted, so range p is a contiguous slice: nothing is packed).
Splitters are quantiles of a sample of every rank's distinct keys (all-gathered, a few KB), so
skewed key distributions still give ranges of equal size; uniform splitters remain available.

The functions below work on torch tensors and a process group only, so the same code runs
under gloo (tests: CPU tensors, also with libgossgpu.so doing the counting of several ranks on
one GPU -- keys are then staged through host memory for the exchange) and under RCCL on GPUs
(bench.py).  One-word keys (2*len <= 62 bits) are int64 tensors of shape [m]: their sign bit is
never set, so signed comparisons order them correctly.  Two-word keys (2*len <= 128) are int64
tensors of shape [m, 2] with columns (lo, hi) -- the library's Key2 layout; both words are
unsigned, so they are compared after flipping the sign bit.

Stream order: the library works on its own HIP stream and waits for no other.  Every tensor
handed to it here is either complete because a collective on it has been waited for
(`_sync`), or was produced by torch kernels that the binding's per-push
`torch.cuda.synchronize()` covers (gossamer_amd.binding._torch_ready); a C caller of the ABI
orders its streams itself (include/goss_gpu.h, goss_gpu_push_run_device).
"""
import torch
import torch.distributed as dist

from .binding import MODE_GRAPH


_SIGN = -(1 << 63)
_MASK = (1 << 64) - 1


def _as_i64(v):
    """unsigned 64-bit value -> the int64 with the same bits"""
    v &= _MASK
    return v - (1 << 64) if v >> 63 else v


def _key_ints(t):
    """python ints of a key tensor ([m] or [m, 2] = (lo, hi))"""
    if t.dim() == 1:
        return [int(x) for x in t.tolist()]
    return [((h & _MASK) << 64) | (l & _MASK) for l, h in t.tolist()]


def _key_tensor(vals, two, device="cpu"):
    if not two:
        return torch.tensor(vals, dtype=torch.int64, device=device).reshape(len(vals))
    rows = [[_as_i64(v), _as_i64(v >> 64)] for v in vals]
    return torch.tensor(rows, dtype=torch.int64, device=device).reshape(len(rows), 2)


def uniform_splitters(key_bits, parts, dtype=torch.int64, device="cpu"):
    """parts-1 interior splitters cutting [0, 2^key_bits) into equal ranges: shape [parts-1] for
    one-word keys (key_bits <= 62), [parts-1, 2] = (lo, hi) for two-word keys.  Hash-chosen
    canonical k-mers of an i.i.d. genome are uniform on the top bits."""
    total = 1 << key_bits
    cuts = [(total * p) // parts for p in range(1, parts)]
    return _key_tensor(cuts, key_bits > 62, device).to(dtype)


def sampled_splitters(sorted_keys, parts, group=None, per_rank=1024):
    """Splitters for skewed data: every rank contributes `per_rank` evenly spaced keys of its
    sorted distinct set, the samples are all-gathered (a few KB) and the p/parts quantiles of
    their union cut the key space.  Every rank computes the same splitters.  A rank with no
    keys contributes nothing; with no keys anywhere the result is all zeros (every key would go
    to the last range -- there are none)."""
    world = dist.get_world_size(group)
    two = sorted_keys.dim() == 2
    m = int(sorted_keys.shape[0])
    width = 2 if two else 1
    sample = torch.zeros((per_rank, width), dtype=torch.int64)
    n = min(m, per_rank)
    if n:
        idx = torch.div(torch.arange(n, dtype=torch.int64) * m, n, rounding_mode="floor").to(sorted_keys.device)
        sample[:n] = sorted_keys.index_select(0, idx).reshape(n, width).cpu()
    xdev = _exchange_device(sorted_keys.device, group)
    mine = torch.cat([torch.tensor([n], dtype=torch.int64), sample.reshape(-1)]).to(xdev)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    allc = torch.stack(allv).cpu()          # (one copy for all ranks' samples)
    if not two:
        # one-word keys are below 2^62: the signed order of the int64 words is theirs (no trip through python ints:
        # 8 x 1 024 of them took 5 ms of a step)
        rows = [allc[r, 1:1 + int(allc[r, 0])] for r in range(world)]
        vals = torch.cat(rows).sort().values if rows else torch.empty(0, dtype=torch.int64)
        n_all = int(vals.numel())
        if not n_all:
            return torch.zeros(parts - 1, dtype=torch.int64, device=sorted_keys.device)
        at = torch.tensor([min(n_all - 1, (n_all * p) // parts) for p in range(1, parts)], dtype=torch.int64)
        return vals.index_select(0, at).to(sorted_keys.device).reshape(parts - 1)
    vals = []
    for r in range(world):
        v = allc[r]
        c = int(v[0].item())
        vals += _key_ints(v[1:].reshape(per_rank, width)[:c])
    vals.sort()
    if not vals:
        cuts = [0] * (parts - 1)
    else:
        cuts = [vals[min(len(vals) - 1, (len(vals) * p) // parts)] for p in range(1, parts)]
    return _key_tensor(cuts, two, sorted_keys.device)


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
        cuts = torch.searchsorted(sorted_keys, splitters.to(sorted_keys.device), right=False).tolist()
    else:
        cuts = [_lower_bound2(sorted_keys, int(lo), int(hi)) for lo, hi in splitters.tolist()]
    edges = [0] + cuts + [n]
    return [edges[i + 1] - edges[i] for i in range(len(edges) - 1)]


def _exchange_device(device, group=None):
    """Where the collectives' buffers live: the GPU under RCCL, host memory under gloo."""
    return torch.device(device) if dist.get_backend(group) == "nccl" else torch.device("cpu")


# RCCL 2.26 (ROCm 7.0, measured on MI355X with one rank sending to itself): an all-to-all segment above 1 GiB arrives
# with its second half missing -- no error.  Every all-to-all below therefore moves at most _A2A_CHUNK bytes per
# (source, destination) pair and round; all ranks derive the number of rounds from sizes they all know.
_A2A_CHUNK = 512 << 20


def all_to_all_views(outs, ins, max_bytes, group=None, async_op=False):
    """all-to-all(v) over 1-D tensors: ins[p] goes to rank p, outs[p] comes from rank p (views of larger buffers are
    fine: nothing is packed).  max_bytes = the largest segment between ANY pair of ranks (the same value on every rank).
    RCCL: list form, in rounds of at most _A2A_CHUNK bytes per segment.  gloo (CPU tensors): one all_to_all_single of
    the concatenation (ProcessGroupGloo has no list form).  async_op (RCCL): returns the rounds' work handles at once
    -- the collective runs on RCCL's stream while the caller goes on; wait for every handle before touching `outs`."""
    if dist.get_backend(group) != "nccl":
        send = torch.cat([t.reshape(-1) for t in ins]) if len(ins) > 1 else ins[0].reshape(-1).contiguous()
        recv = torch.empty(sum(int(t.numel()) for t in outs), dtype=send.dtype)
        dist.all_to_all_single(recv, send, [int(t.numel()) for t in outs], [int(t.numel()) for t in ins], group=group)
        at = 0
        for t in outs:
            n = int(t.numel())
            t.reshape(-1).copy_(recv[at:at + n])
            at += n
        return []
    step = _A2A_CHUNK // ins[0].element_size()
    rounds = max(1, -(-int(max_bytes) // _A2A_CHUNK))
    works = []
    for r in range(rounds):
        w = dist.all_to_all([t[r * step:(r + 1) * step] for t in outs], [t[r * step:(r + 1) * step] for t in ins], group=group,
                            async_op=async_op)
        if async_op:
            works.append(w)
    return works


def exchange_runs(keys, counts, splitters, group=None):
    """all-to-all(v): range p of (keys, counts) goes to rank p.  Returns (recv_keys, recv_counts, recv_sizes) -- the
    concatenation of one sorted run per source rank, and the run lengths.  The keys are sorted, so every range is a
    contiguous slice of both arrays: they are sent as they lie (no packing pass, no staging copy on the GPU)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    send = split_sizes(keys, splitters)
    assert len(send) == world
    # every rank learns the whole matrix of run lengths: one collective, one host synchronisation
    mine = torch.tensor(send, dtype=torch.int64, device=keys.device)
    rows = torch.empty((world, world), dtype=torch.int64, device=keys.device)
    dist.all_gather(list(rows.unbind(0)), mine, group=group)
    matrix = rows.cpu().tolist()
    recv = [int(matrix[q][rank]) for q in range(world)]
    total = sum(recv)
    words = 1 if keys.dim() == 1 else 2
    rk = torch.empty((total,) + tuple(keys.shape[1:]), dtype=torch.int64, device=keys.device)
    rc = torch.empty(total, dtype=torch.int32, device=keys.device)
    largest = max(max(row) for row in matrix)

    def cut(t, sizes, per):          # 1-D views of the slices of t (per elements per item) that go to / come from each rank
        flat, out, at = t.reshape(-1), [], 0
        for n in sizes:
            out.append(flat[at * per:(at + n) * per])
            at += n
        return out
    all_to_all_views(cut(rk, recv, words), cut(keys, send, words), largest * 8 * words, group)
    all_to_all_views(cut(rc, recv, 1), cut(counts, send, 1), largest * 4, group)
    return rk, rc, recv


def gather_counts(m_local, device, group=None):
    """all-gather of the per-range distinct counts -> (list M_p, global M, this rank's offset)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    xdev = _exchange_device(device, group)
    mine = torch.tensor([m_local], dtype=torch.int64, device=xdev)
    allm = [torch.empty(1, dtype=torch.int64, device=xdev) for _ in range(world)]
    dist.all_gather(allm, mine, group=group)
    ms = [int(x) for x in torch.cat(allm).cpu().tolist()]
    return ms, sum(ms), sum(ms[:rank])


def gather_ranges_to_root(keys, counts, ms, group=None):
    """Concatenate every rank's range on rank 0 (ranges are disjoint and ordered by rank, so
    the concatenation is the globally sorted distinct key set)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    xdev = _exchange_device(keys.device, group)
    keys, counts = keys.to(xdev), counts.to(xdev)
    send = [int(keys.shape[0])] + [0] * (world - 1)
    recv = ms if rank == 0 else [0] * world
    rk = torch.empty((sum(recv),) + tuple(keys.shape[1:]), dtype=keys.dtype, device=xdev)
    rc = torch.empty(sum(recv), dtype=counts.dtype, device=xdev)
    words = 1 if keys.dim() == 1 else 2

    def cut(t, sizes, per):
        flat, out, at = t.reshape(-1), [], 0
        for n in sizes:
            out.append(flat[at * per:(at + n) * per])
            at += n
        return out
    all_to_all_views(cut(rk, recv, words), cut(keys, send, words), max(ms) * 8 * words, group)
    all_to_all_views(cut(rc, recv, 1), cut(counts, send, 1), max(ms) * 4, group)
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


def _push_run(ctx, on_gpu, keys_ptr, counts_ptr, n):
    if on_gpu:
        ctx.push_run(keys_ptr, counts_ptr, n)
    else:
        ctx.push_run_host(keys_ptr, counts_ptr, n)


def _push_received(ctx, rk, rc, recv, words):
    """the runs of an exchange (concatenated, on the GPU or in host memory) -> runs of the context"""
    off = 0
    for n in recv:
        if n:
            _push_run(ctx, rk.device.type == "cuda", rk.data_ptr() + off * 8 * words, rc.data_ptr() + off * 4, n)
        off += n


_ROUTE_SIZES = {}          # (bases_ptr, nbytes, parts, k, mode) -> records per part of the last routing of that input (the probe below)


def route_and_exchange_records(ctx, bases_ptr, nbytes, device, group=None, pieces=0):
    """The exchange before counting: this rank's reads -> super-k-mer records routed by minimizer into one buffer
    per rank (goss_gpu_route_records_device) -> all-to-all(v) of record bytes -> the records this rank received are
    counted (goss_gpu_push_records_device; not yet finished).  `pieces` (0 = by size): the reads are routed and
    exchanged in that many pieces, the all-to-all of one overlapping the routing of the next; what arrived is counted
    in one go.  Returns the windows of this rank's own reads."""
    import os
    from .binding import record_bytes
    RB = record_bytes(ctx.k, ctx.mode)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device(device)
    probe = int(os.environ.get("GOSS_DIST_ROUTE_PARTS", "0"))
    if world == 1 and probe >= 1:
        # one rank's load of a `probe`-rank build, measured on one GPU (tools/scale_probe.sh): the records are cut as
        # they would be for that many destinations and the rank takes all its own parts -- the number of windows one
        # rank of the real build receives from everybody
        # (the parts must lie back to back for the ONE push below: sizes that merely suffice -- remembered from another
        # input at the same address, say -- leave gaps between the parts, so the routing is redone until every part takes
        # exactly what it was given; the slots a part takes do not depend on the room it finds)
        memo = (bases_ptr, nbytes, probe, ctx.k, ctx.mode)
        need = _ROUTE_SIZES.get(memo, [1] * probe)          # (exact after the first call on this input)
        for attempt in range(3):
            first = [sum(need[:p]) for p in range(probe)]
            sbuf = torch.empty(max(1, sum(need)) * RB, dtype=torch.uint8, device=dev)
            recs, wins, ok = ctx.route_records(bases_ptr, nbytes, probe, sbuf.data_ptr(), first, need)
            if ok and recs == need:
                break
            need = recs
        else:
            raise RuntimeError("routing did not settle on the sizes it asked for")
        _ROUTE_SIZES[memo] = recs
        ctx.push_records(sbuf.data_ptr(), sum(recs), sum(wins))
        return sum(wins)
    # The reads are cut into pieces (window starts [s_i, s_i+1): piece i = bytes [s_i, s_i+1 + len - 1), so no window is
    # lost or taken twice) and the pieces go through route -> all-to-all as a pipeline: while the records of piece i
    # travel (RCCL's stream), piece i + 1 is routed (the library's stream).
    klen = ctx.k + (1 if ctx.mode == MODE_GRAPH else 0)
    nstarts = nbytes - klen + 1 if nbytes >= klen else 0
    if pieces <= 0:
        pieces = max(1, min(8, nstarts // (2 << 30)))          # ~2 G window starts each: ~4 GB of records per piece
    pieces = max(1, min(pieces, max(1, nstarts // 65536)))
    xdev = _exchange_device(dev, group)
    on_gpu = xdev.type == "cuda"
    if world > 1:          # every piece is a collective: all ranks take the same number (a short share has empty ones)
        agreed = torch.tensor([pieces], dtype=torch.int64, device=xdev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MAX, group=group)
        pieces = int(agreed.item())
    cuts = [nstarts * i // pieces for i in range(pieces + 1)]
    own_windows = 0
    _sync(dev)               # the bases are there; from here on only single streams are waited for
    # What arrives is gathered in TWO buffers -- the first half of the pieces and the rest -- and each is counted in one
    # go: a push is at least one chunk of the counting pipeline, with its sample, its sizing and a run to merge (eight
    # pushes of an eighth each cost ~30 ms more per rank than two chunks of a half), and 125 M reads per rank are two
    # chunks anyway.  Schedule at 8 ranks: the pieces are routed back to back (~5 ms each) and their all-to-alls queue up
    # on RCCL's stream (~8 ms each: the wire is the slower of the two); the first half has landed when the last piece is
    # routed and is counted WHILE the second half travels; only then is the second half waited for.  For that the sizes
    # of a piece (a matrix of 2 x world numbers) must not queue behind the previous piece's records: they go through a
    # group of their own (_meta_group: gloo beside RCCL).
    side = world > 1 and (on_gpu or os.environ.get("GOSS_DIST_META_GROUP") == "1")          # (the variable: for the tests under gloo)
    meta = _meta_group(group) if side else None
    if meta is None:
        meta, side = group, False
    mdev = torch.device("cpu") if side else xdev
    halves = [list(range((pieces + 1) // 2)), list(range((pieces + 1) // 2, pieces))] if pieces >= 2 else [list(range(pieces))]
    gather = [{"buf": None, "cap": 0, "used": 0, "windows": 0, "extra": []} for _ in halves]
    jobs = [None] * pieces       # (works, buffers kept alive)
    depth = 4                    # pieces on their way at once: their send buffers are the memory this loop holds

    def landed(i):
        """piece i has arrived (the host waits for ITS all-to-all, not for the device); its send buffer is let go"""
        if jobs[i] is None:
            return
        works, keep = jobs[i]
        for w in works:
            w.wait()              # (RCCL: the current stream waits for the collective ...)
        if works and dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()          # (... and the host for the current stream)
        jobs[i] = None
        del keep

    last_recs = None
    # (fault injection for the tests: this rank dies while piece GOSS_DIST_FAIL_PIECE is on its way -- its peers then
    # wait in a collective for ever, and whoever started the ranks must end the job: bench.py's launcher does)
    # (only with GOSS_DIST_TEST_HOOKS=1 in the environment: a production run never looks at the two variables)
    hooks = os.environ.get("GOSS_DIST_TEST_HOOKS") == "1"
    fail_rank = int(os.environ.get("GOSS_DIST_FAIL_RANK", "-1")) if hooks else -1
    fail_piece = int(os.environ.get("GOSS_DIST_FAIL_PIECE", "0")) if hooks else 0
    for i in range(pieces):
        if rank == fail_rank and i == min(fail_piece, pieces - 1):
            os._exit(7)
        if i >= depth:
            landed(i - depth)
        ptr = bases_ptr + cuts[i]
        n = nbytes - cuts[i] if i == pieces - 1 else min(cuts[i + 1] - cuts[i] + klen - 1, nbytes - cuts[i])
        # room per part: ~6 windows per record at 8 parts (fewer parts cut less often), a third of slack -- after the
        # first piece what that piece took and a twentieth (the pieces are alike); a part that needs more is reported
        # by the library and the routing is redone with exact sizes
        guess = n // 5 // world + n // 15 // world + 4096
        caps = [guess] * world if last_recs is None else [r + r // 20 + 4096 for r in last_recs]
        for attempt in range(2):
            first = [sum(caps[:p]) for p in range(world)]
            sbuf = torch.empty(max(1, sum(caps)) * RB, dtype=torch.uint8, device=dev)
            recs, wins, ok = ctx.route_records(ptr, n, world, sbuf.data_ptr(), first, caps, ready=True)
            if ok:
                break
            del sbuf
            caps = [x + 1 for x in recs]
        else:
            raise RuntimeError("routing did not fit the sizes it asked for")
        own_windows += sum(wins)
        last_recs = recs
        mine = torch.tensor(recs + wins, dtype=torch.int64, device=mdev)
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine, group=meta)
        matrix = torch.stack(rows).cpu().tolist()
        recv = [int(matrix[q][rank]) for q in range(world)]
        recv_windows = sum(int(matrix[q][world + rank]) for q in range(world))
        total = sum(recv)
        parts = [sbuf[first[p] * RB:(first[p] + recs[p]) * RB] for p in range(world)]
        largest = max(max(row[:world]) for row in matrix) * RB
        if not on_gpu:
            parts = [p.cpu() for p in parts]
        h = 0 if i in halves[0] else 1
        g = gather[h]
        if g["buf"] is None:          # sized from the half's first piece: the pieces are alike
            g["cap"] = int(total * len(halves[h]) * 1.08) + 65536 if len(halves[h]) > 1 else max(1, total)
            g["buf"] = torch.empty(g["cap"] * RB, dtype=torch.uint8, device=dev if on_gpu else "cpu")
        if g["used"] + total <= g["cap"]:
            rbuf = g["buf"][g["used"] * RB:(g["used"] + total) * RB]
            g["used"] += total
            g["windows"] += recv_windows
        else:
            rbuf = torch.empty(max(1, total) * RB, dtype=torch.uint8, device=dev if on_gpu else "cpu")
            g["extra"].append((rbuf, total, recv_windows))
        outs, at = [], 0
        for m in recv:
            outs.append(rbuf[at * RB:(at + m) * RB])
            at += m
        # (views of both buffers: nothing is packed or copied on the GPU)
        works = all_to_all_views(outs, parts, largest, group, async_op=on_gpu and pieces > 1)
        jobs[i] = (works, (sbuf, parts, outs, rbuf))
        del sbuf, parts, outs, rbuf
    for h, g in enumerate(gather):
        # the host waits for THIS half only: what is still queued on RCCL's stream travels while the half is counted
        for i in halves[h]:
            landed(i)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        for buf, nrec, nwin in [(g["buf"], g["used"], g["windows"])] + g["extra"]:
            if nrec:
                if not on_gpu:
                    buf = buf[:nrec * RB].to(dev)
                    _sync(dev)
                ctx.push_records(buf.data_ptr(), nrec, nwin, ready=True)
        g["buf"] = None
        g["extra"] = []
    return own_windows


_META_GROUPS = {}


def _meta_group(group):
    """A gloo group with the ranks of `group`, for the few numbers that must not wait behind bulk data queued on
    RCCL's stream (collectives of one communicator run in order).  Made once per group, by all its ranks together."""
    key = id(group) if group is not None else 0
    if key not in _META_GROUPS:
        ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
        try:
            _META_GROUPS[key] = dist.new_group(ranks=ranks, backend="gloo")
        except Exception:          # no gloo here (all ranks of a node fail alike): the sizes queue behind the data
            _META_GROUPS[key] = None
    return _META_GROUPS[key]


def result_views(ctx, words, device):
    """zero-copy views of the context's result: valid until its next emit, reset, push or close"""
    kp, cp, m = ctx.result_ptrs()
    return key_view(kp, m, words, device), device_view(cp, m, torch.int32, device)


def _carry_big_counts(ctx, keys, counts, words, group):
    """Graph mode: a multiplicity of 2^32 - 1 or more is a marker in the u32 array and an exact u64 beside the result
    (goss_gpu_big_counts).  The marker must not be summed: the sender zeroes it in a copy of its counts, every rank
    learns all (key, exact count) pairs (a handful), and the owner of the key's range adds the exact value after the
    merge (_add_big_counts).  Returns (counts to send, [(key, exact)] of all ranks)."""
    world = dist.get_world_size(group)
    if ctx.mode != MODE_GRAPH:          # (a k-mer set stores no counts)
        return counts, []
    big = sorted(ctx.big_counts().items())
    everyone = [None] * world
    dist.all_gather_object(everyone, big, group=group)
    allbig = [kv for lst in everyone for kv in lst]
    if big:
        counts = counts.clone()
        probe = _key_tensor([k for k, _ in big], words == 2, keys.device)
        if words == 1:
            idx = torch.searchsorted(keys, probe)
        else:
            idx = torch.tensor([_lower_bound2(keys, int(lo), int(hi)) for lo, hi in probe.tolist()], dtype=torch.int64, device=keys.device)
        counts[idx] = 0
    return counts, allbig


def _add_big_counts(ctx, allbig, splitters, words, rank):
    """the exact counts of _carry_big_counts for the keys of THIS rank's range, as single-key runs of at most
    2^32 - 2 each: the merge adds them up and keeps the exact sum beside the result again"""
    import numpy as np
    cuts = _key_ints(splitters)
    for k, exact in allbig:
        owner = sum(1 for c in cuts if k >= c)
        if owner != rank:
            continue
        kw = np.array([k & _MASK] if words == 1 else [k & _MASK, k >> 64], dtype=np.uint64)
        while exact > 0:
            piece = min(exact, (1 << 32) - 2)
            cw = np.array([piece], dtype=np.uint32)
            ctx.push_run_host(kw.ctypes.data, cw.ctypes.data, 1)
            exact -= piece


_T = {"on": None, "t": 0.0, "acc": {}}


def _tick(name, device=None):
    """GOSS_DIST_TIMING=1: host wall clock between the named points of a distributed step (the device waited for at each),
    printed by rank 0 at the end of emit_distributed -- where a step's time outside the kernels goes."""
    import os
    import time
    if _T["on"] is None:
        _T["on"] = os.environ.get("GOSS_DIST_TIMING") == "1"
    if not _T["on"]:
        return
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    now = time.perf_counter()
    if name is not None:
        _T["acc"][name] = _T["acc"].get(name, 0.0) + (now - _T["t"]) * 1e3
    _T["t"] = now


def count_range(ctx, bases_ptr, nbytes, key_bits, device, group=None, splitters="sampled", exchange="counted", record_pieces=0):
    """count -> range-partition -> merge of the received runs: this rank's range of the global result stays in the
    Context.  Returns (this rank's windows, key words, the splitters used).
    `exchange`: "counted" (local count of this rank's reads, then the exchange of its distinct pairs) or "records"
    (super-k-mer records routed by minimizer and exchanged BEFORE counting; module docstring).
    `splitters`: "sampled", "uniform", or a tensor from an earlier call (set algebra: every set must be cut at the
    same places)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    _tick(None, device)
    ctx.reset()
    if exchange == "records" and key_bits <= 126:
        windows = route_and_exchange_records(ctx, bases_ptr, nbytes, device, group, record_pieces)
        _tick("route + exchange + pushes", device)
        c = ctx.finish()
    else:
        ctx.push_device(bases_ptr, nbytes)
        c = ctx.finish()
        windows = c.windows
    _tick("finish (first)", device)
    words = c.key_words
    if (key_bits > 62) != (words == 2):
        raise ValueError("key_bits = %d does not match the context's %d-word keys (2*len: len = k, or k+1 for graphs)" % (key_bits, words))
    keys, counts = result_views(ctx, words, device)
    xdev = _exchange_device(device, group)
    if xdev.type == "cpu":
        keys, counts = keys.cpu(), counts.cpu()
    counts, allbig = _carry_big_counts(ctx, keys, counts, words, group)
    _tick("big counts", device)
    if isinstance(splitters, str):
        splitters = sampled_splitters(keys, world, group) if splitters == "sampled" else uniform_splitters(key_bits, world, device=keys.device)
    _tick("splitters", device)
    rk, rc, recv = exchange_runs(keys, counts, splitters, group)
    _sync(device)          # the library runs on its own stream: finish the collectives first
    _tick("exchange of the counted ranges", device)
    del keys, counts
    # merge the received runs: they replace the local result
    ctx.reset()
    _push_received(ctx, rk, rc, recv, words)
    _add_big_counts(ctx, allbig, splitters, words, rank)
    ctx.finish()
    _tick("received runs pushed and merged", device)
    del rk, rc             # the runs were copied into the library's arena
    return windows, words, splitters


def _assemble_on_root(ctx, keys, counts, ms, M, device, group):
    """ranges -> rank 0 -> one run -> on-disk arrays (in HBM) on rank 0"""
    ak, ac = gather_ranges_to_root(keys, counts, ms, group)
    _sync(device)
    del keys, counts        # views of the result: gone with the reset below
    if dist.get_rank(group) == 0:
        ctx.reset()
        if M:
            _push_run(ctx, ak.device.type == "cuda", ak.data_ptr(), ac.data_ptr(), M)
        ctx.finish()
        ctx.emit_device()


def emit_distributed(ctx, device, first_index, total, group=None, estimate=0):
    """Every rank emits its own span of the object, rank 0 the parts that need all ranges.

    On every rank (goss_gpu_emit_part): its slice of the low-bits column files (and, for graphs, of the
    ord0 byte file) -- they belong at element offset `first_index` of the whole file and stay in the
    rank's HBM for its own writer -- and its SPAN of the high-bits bitmap, built from its own keys (".part.span":
    about 2.4 bits per key) with, behind it, the DenseSelect blocks of "-d0" / "-d1" that lie wholly inside the rank's
    share of the zeros / ones.  Only those and the few records of counts > 255 / the count histogram travel to
    rank 0, which ORs the spans together (neighbours share their boundary words), builds the blocks that straddle two
    ranks from the assembled bitmap, writes the header, master index and rank array of -d0 / -d1 and, for graphs, ord1 / ord2 with their presence arrays and the histogram text
    (goss_gpu_emit_assemble).  Returns {suffix: (size, device address)} of this rank's files."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # where the ranges below this one end (the high part of their last key): which zeros of the bitmap are this range's,
    # so that it builds its own blocks of "-d0" as well as of "-d1" (goss_gpu_emit_part_ranges) -- P numbers gathered
    _tick("sizes gathered", device)
    high, nonempty = ctx.emit_last_high(total, estimate)
    mine_h = torch.tensor([high if high < (1 << 63) else high - (1 << 64), 1 if nonempty else 0], dtype=torch.int64, device=_exchange_device(device, group))
    all_h = [torch.empty_like(mine_h) for _ in range(world)]
    dist.all_gather(all_h, mine_h, group=group)
    prev = 0
    for r in range(rank):
        h, ne = (int(x) for x in all_h[r].cpu().tolist())
        if ne:
            prev = h & ((1 << 64) - 1)
    _tick("last high parts gathered", device)
    ctx.emit_part(first_index, total, estimate, prev_last_high=prev)
    _tick("own part emitted", device)
    files = {name: (size, ptr) for name, size, ptr in ctx.file_list()}
    size, ptr = files[".part.span"]
    xdev = _exchange_device(device, group)
    mine = device_view(ptr, size, torch.uint8, device).to(xdev) if size else torch.empty(0, dtype=torch.uint8, device=xdev)
    # sizes of every rank's parts: one small all-gather
    small = b""
    if ".part.big" in files:
        small = ctx.read_file(".part.big")
    hist = ctx.read_file(".part.hist") if ".part.hist" in files else b""
    meta = torch.tensor([size, len(small), len(hist)], dtype=torch.int64, device=xdev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu().tolist()
    # span + small records -> rank 0: ONE all-to-all in which only rank 0 receives (every rank's bytes = its span, its
    # counts above 255 and its histogram, back to back; rank 0 knows the three sizes of every rank from `metas`)
    rec = torch.frombuffer(bytearray(small + hist), dtype=torch.uint8).to(xdev) if small or hist else torch.empty(0, dtype=torch.uint8, device=xdev)
    both = torch.cat([mine, rec]) if rec.numel() else mine
    per = [int(r[0] + r[1] + r[2]) for r in metas]
    recv = per if rank == 0 else [0] * world
    send = [int(both.numel())] + [0] * (world - 1)
    allb = torch.empty(max(1, sum(recv)), dtype=torch.uint8, device=xdev)

    def cut(t, sizes):
        out, at = [], 0
        for n in sizes:
            out.append(t[at:at + n])
            at += n
        return out
    all_to_all_views(cut(allb, recv), cut(both, send), max(per), group)
    _sync(device)
    _tick("spans to rank 0", device)
    if rank == 0:
        spans, big, hst, at = [], b"", b"", 0
        for r in metas:
            n0, n1, n2 = int(r[0]), int(r[1]), int(r[2])
            spans.append(allb[at:at + n0])
            if n1 or n2:
                raw = bytes(allb[at + n0:at + n0 + n1 + n2].cpu().numpy().tobytes())
                big += raw[:n1]
                hst += raw[n1:]
            at += n0 + n1 + n2
        # (the spans side by side, as the assembler takes them; with one rank, or no records between them, they already are)
        allhigh = spans[0] if world == 1 else torch.cat(spans)
        hp = allhigh.to(device) if allhigh.device.type != "cuda" else allhigh
        ctx.emit_assemble(hp.data_ptr(), sum(int(r[0]) for r in metas), total, estimate, big, hst)
        del hp
    _tick("assembled", device)
    if _T["on"] and rank == 0:
        import sys
        sys.stderr.write("dist timing (ms, summed over the steps so far): " + ", ".join("%s %.2f" % kv for kv in _T["acc"].items()) + "\n")
    return {name: (size, ptr) for name, size, ptr in ctx.file_list()}


def count_distributed(ctx, bases_ptr, nbytes, key_bits, device, group=None, emit_on_root=True, splitters="sampled",
                      emission="distributed", exchange="counted", record_pieces=0):
    """The whole multi-GPU job on an already created Context (either mode, one- or two-word keys):
    local count -> exchange -> merge own range -> all-gather M -> emit.  emission = "distributed": every
    rank emits its span, rank 0 the index (emit_distributed); "root": the ranges are gathered on rank 0,
    which builds every file (the first form of this path, kept for comparison).
    Returns dict(windows=<this rank's windows>, M=<global distinct>, m_range=<this range>, ranges, first)."""
    windows, words, _ = count_range(ctx, bases_ptr, nbytes, key_bits, device, group, splitters, exchange, record_pieces)
    m_range = ctx.counts.distinct
    ms, M, first = gather_counts(m_range, device, group)
    out = {"windows": windows, "M": M, "m_range": m_range, "ranges": ms, "first": first}
    if emit_on_root:
        if emission == "distributed":
            out["files"] = emit_distributed(ctx, device, first, M, group)
        else:
            keys, counts = result_views(ctx, words, device)
            _assemble_on_root(ctx, keys, counts, ms, M, device, group)
    return out


def assemble_files(per_rank_files):
    """{suffix: bytes} of the whole object from every rank's {suffix: bytes} in rank order: the slices
    (low-bits columns, ord0) concatenate; everything else comes from rank 0; the .part.* transport files
    are dropped.  What a set of per-rank writers produces by writing each slice at its element offset."""
    out = {}
    for name, data in per_rank_files[0].items():
        if name.startswith(".part."):
            continue
        if ".low-bits" in name and ".ord" not in name or name == "-counts.ord0":
            out[name] = b"".join(f.get(name, b"") for f in per_rank_files)
        else:
            out[name] = data
    return out


def set_algebra_distributed(ctx, inputs, key_bits, op, device, group=None, emit_on_root=True, exchange="counted"):
    """BASELINE config C5 across ranks: every input (bases_ptr, nbytes) is this rank's share of the
    reads of one k-mer set.  The sets are counted and range-partitioned one after the other with
    the same splitters (those sampled from the first set), so the set operation needs no further
    exchange: each rank combines its ranges (goss_gpu_select_counts on weighted runs, as the
    single-GPU commands do), rank 0 assembles the result.  op = "intersect" (all sets; globally
    empty ones are skipped, as GossCmdIntersectKmerSets.cc:29-79 does) or "subtract" (first minus
    second, GossCmdSubtractKmerSet.cc:47-66).  Returns dict(sizes=<global size of every input>,
    M=<result>)."""
    if op not in ("intersect", "subtract") or (op == "subtract" and len(inputs) != 2):
        raise ValueError("op must be 'intersect' or 'subtract' (exactly two sets)")
    ranges, sizes, words = [], [], 1
    splitters = "sampled"
    for ptr, nbytes in inputs:
        _, words, splitters = count_range(ctx, ptr, nbytes, key_bits, device, group, splitters, exchange)
        keys, _ = result_views(ctx, words, device)
        ranges.append(keys.clone())          # the context is reused for the next set
        sizes.append(gather_counts(ranges[-1].shape[0], device, group)[1])
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
        keys = ranges[i]
        if keys.shape[0]:
            held.append(torch.full((keys.shape[0],), weights[i], dtype=torch.int32, device=device))
    _sync(device)          # the weights are written on torch's stream, the library copies on its own
    for keys, w in zip([ranges[i] for i in use if ranges[i].shape[0]], held):
        ctx.push_run(keys.data_ptr(), w.data_ptr(), keys.shape[0])
    ctx.finish()
    ctx.select_counts(keep, keep)
    m = int(ctx.result_ptrs()[2])
    ms, M, first = gather_counts(m, device, group)
    if emit_on_root:
        emit_distributed(ctx, device, first, M, group)
    return {"sizes": sizes, "M": M}
