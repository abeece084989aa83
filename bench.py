#!/usr/bin/env python3
"""bench.py -- M canonical k-mers/s counted (k=25, 150 bp reads) on N MI355X GPUs.

A "step" is one pass of the hot path over the whole synthetic read set that is already
resident in HBM: 2-bit/rolling canonical k-mer extraction -> radix sort -> run compaction ->
SparseArray (Elias-Fano + DenseSelect) images of the KmerSet, all on the device.  For N > 1
every rank counts its own 100 M reads (weak scaling), the sorted key space is range-partitioned
with one RCCL all-to-all(v), and rank 0 assembles and emits the object.

Prints ONE JSON line on rank 0 (contract in the task description).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_window(k, read_len, key_bytes=8, keys_per_window=1):
    """packed read traffic + one write and one read of each key."""
    return read_len / (read_len - k + 1) * 3.0 / 8.0 + 2.0 * key_bytes * keys_per_window


def kernels_hash():
    """sha256 over the sources of the kernels profiles/traffic.json holds figures for (extraction, partition, counting):
    their code, not their comments or layout (the kernel sources hold no string literal with comment marks)."""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gossamer_amd", "csrc")
    for f in [os.path.join(d, n) for n in ("goss_key.hpp", "kernels_common.hpp", "kernels_count.hpp", "kernels_extract.hpp", "kernels_partition.hpp")]:
        with open(f, "r", encoding="utf-8") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        text = " ".join(text.split())
        h.update(os.path.basename(f).encode() + b"\0" + text.encode())
    return h.hexdigest()[:16]


def measured_traffic(kernel_class):
    """HBM bytes per unit of the dominant kernel from the PMC passes committed under profiles/
    ((2*FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md HBM section).  The table is keyed by a hash of
    the kernel sources it was measured on: returns (entry or None, warning or None)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/traffic.json missing"
    if t.get("kernels_sha") != kernels_hash():
        return None, ("profiles/traffic.json was measured on kernel sources %s, the tree has %s: re-run tools/profile_round.sh"
                      % (t.get("kernels_sha"), kernels_hash()))
    return t.get(kernel_class), None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def note(msg):
    """Progress on stderr (the JSON line is the only thing on stdout): were a side record ever to hang, the last note says which."""
    sys.stderr.write("bench.py [%s] %s\n" % (time.strftime("%H:%M:%S"), msg))
    sys.stderr.flush()


def run_bounded(cmd, seconds, **kw):
    """subprocess.run with a time limit: a child that does not end is killed and reported, never waited for -- the headline
    line must not be lost to a side record (the end-to-end CLI, the probes, the CPU baseline all run as children)."""
    import subprocess
    try:
        return subprocess.run(cmd, timeout=seconds, **kw), None
    except subprocess.TimeoutExpired:
        return None, "timed out after %d s: %s" % (seconds, " ".join(str(c) for c in cmd[:4]))


def cpu_baseline_child(k, read_len, genome_len, seed, sample_reads):
    """cpu_baseline() in a child process with a time limit (the oracle's threaded driver runs 64 threads of C)."""
    import subprocess
    p, err = run_bounded([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "-k", str(k), "--read-len", str(read_len),
                          "--genome", str(genome_len), "--seed", str(seed), "--cpu-sample-reads", str(sample_reads)], 420,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if err:
        return {"failed": err}
    try:
        return json.loads(p.stdout.decode().strip().splitlines()[-1])
    except (ValueError, IndexError):
        return {"failed": p.stderr.decode(errors="replace")[-300:]}


def cpu_baseline(k, read_len, genome_len, seed, sample_reads):
    """The CPU oracle (oracle/, a restatement of the reference algorithm: kind "port") timed on
    bounded samples of the same synthetic workload: on one core, and with T = min(cores, 64) worker
    threads in the thread structure of the reference (GossCmdBuildKmerSet.tcc:226-256: T consumers,
    T sort workers, serial flush).  The reported value is the T-thread figure."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gossamer_amd as g
    import oracle_lib as o
    o.lib()
    cores = os.cpu_count() or 1
    T = max(1, min(cores, 64))
    reads1 = g.synth_reads_host(sample_reads, read_len, genome_len, seed=seed)
    t0 = time.perf_counter()
    files, nwin1 = o.build_kmer_set([(o.LINE, "reads", reads1)], k)
    dt1 = time.perf_counter() - t0
    one = {"value": nwin1 / dt1 / 1e6, "unit": "M k-mers/s", "cores": 1,
           "sample": "first %d reads (%d k-mers) through oracle go_build_kmer_set, %.1f s" % (sample_reads, nwin1, dt1)}
    del files
    # the threaded run gets a sample that grows with the cores, bounded by a quarter of the host's
    # memory (40 bytes per k-mer while sorting) and by 6 M reads
    try:
        ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    except (ValueError, OSError):
        ram = 64 << 30
    kmers_per_read = read_len - k + 1
    nT = int(min(6_000_000, sample_reads // 4 * T, ram / 4 / 40 / kmers_per_read))
    nT = max(nT, sample_reads // 4)
    readsT = reads1 if nT == sample_reads else g.synth_reads_host(nT, read_len, genome_len, seed=seed)
    t0 = time.perf_counter()
    files, nwinT = o.build_kmer_set_mt(readsT, k, T)
    dtT = time.perf_counter() - t0
    eight = None
    if T > 8:          # the thread count of BASELINE.md section 2's figure for the reference's own code (7.45 M k-mers/s)
        n8 = min(nT, sample_reads * 2)
        t0 = time.perf_counter()
        _, nwin8 = o.build_kmer_set_mt(readsT[:n8 * (read_len + 1)] if n8 < nT else readsT, k, 8)
        dt8 = time.perf_counter() - t0
        eight = {"value": nwin8 / dt8 / 1e6, "unit": "M k-mers/s", "cores": 8,
                 "sample": "first %d reads, go_build_kmer_set_mt with 8 threads, %.1f s; BASELINE.md section 2 has the "
                           "reference's own code at 7.45 M k-mers/s with -T 8 (other hardware)" % (n8, dt8)}
    return {"value": nwinT / dtT / 1e6, "unit": "M k-mers/s", "cores": T, "kind": "port", "cpu_model": cpu_model(),
            "eight_threads": eight,
            "host_cores": cores,
            "sample": "first %d reads of the same synthetic set (%d k-mers) through oracle go_build_kmer_set_mt with %d "
                      "threads (per shard: parse + canonicalise + sort-count; parallel merge by key range; serial "
                      "KmerSet emit, in memory), %.1f s" % (nT, nwinT, T, dtT),
            "one_core": one}


def e2e_record(nreads, read_len, genome_len, seed, k, threads):
    """SURVEY.md section 8(d)'s metric: FASTQ file -> `goss build-kmer-set -T n` -> KmerSet files closed, wall clock of
    the command (process start-up, parsing, 2-bit packing on the parser threads, PCIe, HBM mapping, counting, emit,
    file writes included), in a fresh process.  The file holds the bench's own read set (goss synth-reads: the same
    generator, genome and seed), C2's 100 M reads when /dev/shm has room for them (31.5 GB), else as many as fit."""
    import re
    import shutil
    import subprocess
    import tempfile
    goss = os.path.join(ROOT, "gossamer_amd", "goss")
    per_read = 2 * read_len + 16
    base, want = None, nreads
    for cand in ("/dev/shm", tempfile.gettempdir(), ROOT):
        try:
            st = os.statvfs(cand)
            room = st.f_bavail * st.f_frsize - (6 << 30)          # the object's files and some air
            if os.access(cand, os.W_OK) and room > 2_000_000 * per_read:
                base, want = cand, int(min(nreads, room // per_read))
                break
        except OSError:
            continue
    if base is None or not os.path.exists(goss):
        return {"skipped": "no room for a FASTQ sample or no goss executable"}
    d = tempfile.mkdtemp(prefix="goss_e2e_", dir=base)
    try:
        fq = os.path.join(d, "reads.fq")
        t0 = time.perf_counter()
        p, err = run_bounded([goss, "synth-reads", str(want), str(read_len), str(genome_len), str(seed), fq], 300, stderr=subprocess.PIPE)
        if err:
            return {"failed": err}
        if p.returncode != 0:
            return {"failed": "synth-reads: " + p.stderr.decode(errors="replace")[-300:]}
        gen_s = time.perf_counter() - t0
        nbytes = os.path.getsize(fq)
        runs, logs = [], []
        for run_no in range(2):          # (the first run of a freshly written file maps cold pages)
            for f in os.listdir(d):
                if f.startswith("ks"):
                    os.unlink(os.path.join(d, f))
            if run_no:
                # (the process before has ended, the kernel is still giving its 24 GB of HBM and its locked pages back:
                # a build started at once shares the driver with that -- its context takes 0.2 s instead of 0.07)
                time.sleep(2.0)
            t0 = time.perf_counter()
            p, err = run_bounded([goss, "build-kmer-set", "-k", str(k), "-T", str(threads), "-i", fq, "-O", os.path.join(d, "ks"), "-v"], 150,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if err:
                return {"failed": err}
            runs.append(time.perf_counter() - t0)
            logs.append(p.stderr.decode(errors="replace"))
            if p.returncode != 0:
                return {"failed": logs[-1][-400:]}
        secs = min(runs)
        log = logs[runs.index(secs)]

        def phases(wall, text):
            """where a run's wall clock went, from goss -v's own time stamps (seconds since the command object started)"""
            def at(pat):
                mm = re.search(pat, text)
                return float(mm.group(1)) if mm else None
            ready, parsed, merged = at(r"contexts ready at ([0-9.eE+-]+)s"), at(r"reads at ([0-9.eE+-]+)s"), at(r"merged at ([0-9.eE+-]+)s")
            written, total = at(r"written at ([0-9.eE+-]+)s"), at(r"total build time: ([0-9.eE+-]+)s")
            arena = at(r"GB mapped in ([0-9.eE+-]+)s")
            if None in (ready, parsed, merged, written, total):
                return None
            if "beside the parser" in text:
                # (the context is created by a thread of its own while the parser's workers already read and frame)
                return {"process_start_and_exit": round(wall - total, 3), "parse_and_push": round(parsed, 3),
                        "context_beside_the_parser": round(ready, 3),
                        "finish": round(merged - parsed, 3), "emit_and_write": round(written - merged, 3),
                        "arena_mapping_beside_the_parser": arena}
            return {"process_start_and_exit": round(wall - total, 3), "context": round(ready, 3), "parse_and_push": round(parsed - ready, 3),
                    "finish": round(merged - parsed, 3), "emit_and_write": round(written - merged, 3),
                    "arena_mapping_beside_the_parser": arena}
        # the parser alone: the parallel FASTQ framer the build uses, bases written to /dev/null (after the builds: the
        # first pass over a freshly written file maps cold pages; the faster of two)
        parse = []
        for _ in range(2):
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as null:
                _, err = run_bounded([goss, "dump-bases", "-T", str(threads), "-i", fq], 90, stdout=null, stderr=subprocess.PIPE)
            if err:
                break
            parse.append(time.perf_counter() - t0)
        parse_s = min(parse) if parse else None
        m = re.search(r"HBM arena: (\d+) GB mapped in ([0-9.]+)s", log)
        w = re.search(r"k-mer windows: (\d+)", log)
        out_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("ks"))
        windows = int(w.group(1)) if w else want * (read_len - k + 1)
        return {"what": "goss build-kmer-set -k %d -T %d on a %d-read 4-line FASTQ file in %s (the bench's read set): process start -> "
                        "KmerSet files closed; the faster of two runs, two seconds apart" % (k, threads, want, base),
                "reads": want, "fastq_bytes": nbytes, "seconds": secs, "runs_seconds": runs, "value": windows / secs / 1e6,
                "first_run_seconds": runs[0],
                "phases": phases(secs, log), "first_run_phases": phases(runs[0], logs[0]),
                "phases_what": "seconds by goss -v's stamps: process start + runtime load + exit (wall - total build time), parse + pack + "
                               "push loop (the GPU context is created beside it: ready at context_beside_the_parser, not a term of "
                               "the sum), finish (last chunks counted, runs merged, canonical order), emit + file writes; the "
                               "arena is mapped by a thread of its own beside the parser",
                "unit": "M k-mers/s", "parse_only_seconds": parse_s, "parser_GB_per_s": (nbytes / parse_s / 1e9) if parse_s else None,
                "parse_only_what": "goss dump-bases -T %d (the build's parallel FASTQ framer, bases to /dev/null)" % threads,
                "generate_seconds": gen_s, "arena_GB": int(m.group(1)) if m else None,
                "arena_map_ms": float(m.group(2)) * 1e3 if m else None, "output_bytes": out_bytes}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def multi_gpu_probe(ranks=8):
    """No multi-GPU node may be at hand: ONE rank's share of a `ranks`-GPU build (C3's share: 125 M reads per GPU),
    measured on this GPU by child runs of this script (tools/scale_probe.sh does the same for 1 / 2 / 4 / 8):
      counted_1   the rank alone: the multi-GPU code path with one rank (its N = 1 step);
      records_N   the rank routes its reads into super-k-mer records as for N destinations and counts all its own
                  parts: the windows (15.7 G) and distinct keys (125 M) one rank of the real build receives.
    The projection adds the wire and the second exchange by arithmetic: it is not a measurement."""
    import subprocess
    common = ["--force-dist", "--reads", "125000000", "--genome", "125000000", "--steps", "2", "--warmup", "1", "--no-extra",
              "--no-cpu-baseline", "--e2e-reads", "0"]
    out = {"what": "one rank's share of a %d-GPU build (125 M reads per GPU) measured on one GPU; projection by arithmetic" % ranks}
    for name, extra in (("counted_1", ["--exchange", "counted"]), ("records_%d" % ranks, ["--exchange", "records", "--route-parts", str(ranks)])):
        p, err = run_bounded([sys.executable, os.path.abspath(__file__)] + common + extra, 300, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if err:
            out[name] = {"failed": err}
            continue
        try:
            d = json.loads(p.stdout.decode().strip().splitlines()[-1])
            out[name] = {"ms_per_step": d["ms_per_step"], "windows_per_step": d["value"] * 1e6 * d["ms_per_step"] * 1e-3,
                         "device_ms_per_step": d["roofline"]["device_ms_per_step"], "assemble_ms": d["roofline"].get("assemble_ms")}
        except (ValueError, IndexError, KeyError):
            out[name] = {"failed": p.stderr.decode(errors="replace")[-300:]}
    try:
        rec, one = out["records_%d" % ranks], out["counted_1"]
        bytes_per_window = 1.6          # measured: 7.5 windows per 12-byte record at 8 destinations
        wire_gb = rec["windows_per_step"] * bytes_per_window * (ranks - 1) / ranks / 1e9
        wire_ms = wire_gb / (ranks - 1) / 50.0 * 1e3          # one xGMI link per peer, 50 GB/s of its 76.8 achieved
        # gossamer_amd/dist.py: the pieces are routed back to back, their all-to-alls queue on RCCL's stream, and the
        # first half of the records is counted while the second half travels: the wire runs beside device work unless
        # it is longer than half of the step; while it is busy RCCL's copy kernels hold about a tenth of the CUs
        exposed = max(0.0, wire_ms - 0.5 * rec["ms_per_step"]) + 0.1 * wire_ms + 10.0          # + the second exchange and merge
        # emission: every rank builds its own slices, its span of the high-bits bitmap and (round 5) the DenseSelect blocks
        # inside its own ones and zeros -- all inside the measured step ("emit"); rank 0 then takes the spans and blocks
        # (2.44 + ~0.3 bits per distinct key of the whole build, ranks - 1 links at once), ORs the spans, counts the ones per
        # word and composes -d0 / -d1: the measured goss_gpu_emit_assemble of this rank's range (which it does for its own
        # range inside the step already) once more for every OTHER rank's worth of keys
        distinct_all = ranks * 125e6
        span_ms = distinct_all * 2.75 / 8 / 1e9 / ((ranks - 1) * 50.0) * 1e3
        asm_1 = rec.get("assemble_ms") or 0.75 * rec["device_ms_per_step"].get("emit", 0.0)
        emission = span_ms + asm_1 * (ranks - 1)
        step = rec["ms_per_step"] + exposed + emission
        out["projection"] = {"per_rank_step_ms": step, "ratio_to_one_rank": step / one["ms_per_step"],
                             "emission_ms": emission,
                             "value_M_kmers_per_s": ranks * rec["windows_per_step"] / (step * 1e-3) / 1e6,
                             "assumed": "records on the wire %.1f GB per rank over %d links at 50 GB/s = %.0f ms, beside the routing of "
                                        "the later pieces and the counting of the first half of the records (exposed: what exceeds "
                                        "half of the step, here %.0f ms; a tenth of the wire time for RCCL's kernels on the CUs); "
                                        "second exchange and merge 10 ms; emission on rank 0: spans and DenseSelect blocks of the other ranks in %.1f ms, "
                                        "assembling them (bitmap ORed, ones counted, blocks that straddle two ranks built, index written) at "
                                        "the %.2f ms this rank's own range took, per other rank, for all %.0f M keys"
                                        % (wire_gb, ranks - 1, wire_ms, max(0.0, wire_ms - 0.5 * rec["ms_per_step"]), span_ms, asm_1, distinct_all / 1e6)}
    except KeyError:
        pass
    return out


def c4_record(g, torch, device, local_rank):
    """BASELINE config C4: build-graph k = 55 (112-bit edge keys, both strands), 200 M x 150 bp reads of a
    100 Mbp genome, one GPU, Graph emitted; one warm-up and one timed build, inputs resident in HBM."""
    k, L, n, G = 55, 150, 200_000_000, 100_000_000
    try:
        bases = torch.empty(n * (L + 1), dtype=torch.uint8, device=device)
        free_b, _ = torch.cuda.mem_get_info(device)
        ctx = g.Context(k, g.MODE_GRAPH, device=local_rank, hbm_budget=int(free_b * 0.94))
        ctx.synth_reads(bases.data_ptr(), n, L, G, seed=1)
        torch.cuda.synchronize(device)
        ms, c = [], None
        chunks_before = 0
        for it in range(2):
            ctx.reset()
            ctx.timing(reset=True)
            chunks_before = ctx.stat("fused_chunks")          # (the counter runs on over a reset: the timed build's share is the difference)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            ctx.push_device(bases.data_ptr(), bases.numel())
            c = ctx.finish()
            ctx.emit_device()
            torch.cuda.synchronize(device)
            ms.append((time.perf_counter() - t0) * 1e3)
        tim = ctx.timing().as_dict()
        b = algorithmic_bytes_per_window(k + 1, L, 16, 2)
        rec = {"workload": "C4: build-graph k=55, %d x %d bp synthetic reads (genome %d bp, seed 1), Graph emitted" % (n, L, G),
               "ms": ms[-1], "warmup_ms": ms[0], "windows": c.windows, "keys": c.keys, "distinct_edges": c.distinct,
               "windows_per_s": c.windows / (ms[-1] * 1e-3), "value": c.windows / (ms[-1] * 1e-3) / 1e6,
               "unit": "M rho-mer windows/s", "dtype": "u128",
               "roofline": {"bound": "hbm", "algorithmic_bytes_per_window": b, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "achieved": b * c.windows / (ms[-1] * 1e-3) / 1e9,
                            "frac": b * c.windows / (ms[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "device_ms": {nm: v["ms"] for nm, v in tim.items()}},
               "chunks": ctx.stat("fused_chunks") - chunks_before}
        ctx.close()
        return rec
    except Exception as e:          # the headline line must not be lost to the side record
        return {"failed": repr(e)[:300]}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without an outer launcher: this process starts the N rank processes (one
    per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and does nothing else -- it never
    imports torch, never loads libgossgpu.so and never touches a GPU, so nothing is exec'ed or forked from a
    process that has initialised HIP.  Rank 0's standard output (the JSON line) is relayed last; the exit
    code is non-zero if any rank's is."""
    import subprocess
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    # (a rank that never ends -- a collective nobody answers -- must not hold the launcher for ever: after
    # GOSS_BENCH_RANK_TIMEOUT seconds, 1 800 by default, the ranks are ended and the launcher fails)
    deadline = time.time() + float(os.environ.get("GOSS_BENCH_RANK_TIMEOUT", "1800"))
    first_term = None          # when the ranks were first told to end: ten seconds later whoever is left is killed
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes) or time.time() > deadline:
            if first_term is None:
                first_term = time.time()
            for r, p in enumerate(procs):          # a rank failed: its peers would wait in a collective for ever
                if codes[r] is None:
                    p.terminate()
            if time.time() > first_term + 10:      # (a rank inside a collective does not see SIGTERM)
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.kill()
                        codes[r] = -9
        time.sleep(0.05)
    reader.join(10)
    out0 = b"".join(chunks)
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    rc = next((c for c in codes if c != 0), 0)
    if rc:
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
    raise SystemExit(1 if rc else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process group of the N > 1 path: nccl (= RCCL over xGMI, one GPU per rank) or gloo (the exchange "
                         "staged through host memory; ranks share GPUs round robin when there are fewer GPUs than ranks)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: 100 M on one GPU = BASELINE config C2; "
                    "125 M with several GPUs = C3's 10^9 reads on 8 GPUs)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=0, help="genome length per GPU's worth of reads (default: 100 Mbp on one GPU; "
                    "125 Mbp per GPU with several = C3's 1 Gbp genome on 8 GPUs)")
    ap.add_argument("-k", type=int, default=25)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample-reads", type=int, default=1_500_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) time the CPU oracle and print its record: the child "
                    "process of the default run's cpu_baseline")
    ap.add_argument("--e2e-reads", type=int, default=100_000_000, help="reads of the end-to-end CLI record: C2's 100 M by default, "
                    "fewer when the scratch file system has no room (0 = skip)")
    ap.add_argument("--no-packed", action="store_true", help="skip the side record of the same workload with the reads resident in HBM in the packed form")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed-by-the-headline C4 record (build-graph k=55, 200 M reads)")
    ap.add_argument("--hbm-budget-gb", type=float, default=0.0)
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path even with one rank")
    ap.add_argument("--exchange", default="auto", choices=("auto", "records", "counted"),
                    help="N > 1: what travels in the first all-to-all -- super-k-mer records routed by minimizer BEFORE counting "
                         "(each rank counts 1/N of the key space), or the (key,count) pairs of every rank's local count; auto: "
                         "records from 4 ranks on (with 2 ranks one xGMI link would carry half of all records, and counting "
                         "first is cheaper anyway: tools/scale_probe.sh)")
    ap.add_argument("--route-parts", type=int, default=0, help="with --force-dist on one rank: cut the records as a build over "
                    "this many ranks would (the rank then receives all its own parts: one rank's load of an N-rank build)")
    ap.add_argument("--graph", action="store_true", help="build-graph instead of build-kmer-set (windows are (k+1)-mers, "
                    "two keys per window); not the headline metric")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.k, args.read_len, args.genome, args.seed, args.cpu_sample_reads)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])          # does not return
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE = %d but --gpus %d" % (world, args.gpus))
    if args.exchange == "auto":
        args.exchange = "records" if world >= 4 or args.route_parts >= 4 else "counted"

    import torch
    import torch.distributed as dist
    import gossamer_amd as g
    from gossamer_amd import dist as gdist

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py: no GPU visible (the product has no CPU path)")
    if args.backend == "nccl" and ndev < world:
        raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, %d visible (use --backend gloo to share GPUs)" % (world, world, ndev))
    dev_index = local_rank % ndev
    sharing = (world + ndev - 1) // ndev          # ranks per GPU (1 except under gloo on a small box)
    # (the pool's boxes end a run with more than six processes on one card: refused here, before any rank has opened
    # the GPU -- counting devices does not -- so that every rank exits and the launcher reports it)
    max_sharing = int(os.environ.get("GOSS_BENCH_MAX_RANKS_PER_GPU", "4"))
    if sharing > max_sharing:
        raise SystemExit("bench.py: %d ranks on %d GPU(s) = %d processes per card, at most %d are allowed (--backend gloo shares "
                         "GPUs for tests: use --gpus %d or fewer here; GOSS_BENCH_MAX_RANKS_PER_GPU raises the bound)"
                         % (world, ndev, sharing, max_sharing, max_sharing * ndev))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if args.route_parts:
        os.environ["GOSS_DIST_ROUTE_PARTS"] = str(args.route_parts)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    xdev = device if args.backend == "nccl" else torch.device("cpu")          # where small collectives' tensors live

    k, L = args.k, args.read_len
    # BASELINE.json configs: C2 on one GPU; with N GPUs the per-GPU share of C3 (10^9 reads of a
    # 1 Gbp genome on 8 GPUs), so that --gpus 8 IS C3 and per-GPU work is fixed as N grows
    nreads = args.reads or (100_000_000 if world == 1 else 125_000_000)
    genome_len = (args.genome or (100_000_000 if world == 1 else 125_000_000)) * world
    if args.graph:
        config_name = "build-graph"
    elif args.reads or args.genome:
        config_name = "custom"
    else:
        config_name = "C2" if world == 1 else ("C3" if world == 8 else "C3 share per GPU")
    nbytes = nreads * (L + 1)

    # ---- synthetic input, generated on the device (untimed) ------------------------------
    bases = torch.empty(nbytes, dtype=torch.uint8, device=device)
    free_b, total_b = torch.cuda.mem_get_info(device)
    # the exchange buffers of the multi-GPU path are torch tensors outside the library's arena
    # (received runs, this rank's range, the gathered set on rank 0: ~21 GB at 8 ranks)
    budget = int(free_b * 0.94)
    if use_dist:
        # (the record exchange holds this rank's records twice for a moment: as routed and as received, ~1.6 bytes per input byte each)
        budget = min(budget, free_b - max(32 << 30, int(3.6 * nbytes) if args.exchange == "records" else 0))
    if sharing > 1:          # ranks that share a GPU split it (their mem_get_info calls race with each other's allocations)
        budget = int(total_b * 0.8) // sharing - nbytes
    if args.hbm_budget_gb > 0:
        budget = int(args.hbm_budget_gb * (1 << 30))
    ctx = g.Context(k, g.MODE_GRAPH if args.graph else g.MODE_KMER_SET, device=dev_index, hbm_budget=budget)
    ctx.synth_reads(bases.data_ptr(), nreads, L, genome_len, seed=args.seed, first_read=rank * nreads)
    torch.cuda.synchronize(device)

    def step():
        if not use_dist:
            ctx.reset()
            ctx.push_device(bases.data_ptr(), nbytes)
            c = ctx.finish()
            ctx.emit_device()
            return c.windows, c.distinct
        r = gdist.count_distributed(ctx, bases.data_ptr(), nbytes, 2 * (k + 1 if args.graph else k), device, exchange=args.exchange)
        return r["windows"], r["M"]

    def barrier():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    ctx.timing(reset=True)
    barrier()
    t0 = time.perf_counter()
    windows = distinct = 0
    for _ in range(args.steps):
        w, distinct = step()
        windows += w
    barrier()
    dt = time.perf_counter() - t0
    tim = ctx.timing().as_dict()
    w_local = windows
    fused = ctx.stat("fused_chunks") > 0
    narrow = ctx.stat("narrow_chunks") > 0        # ... and reads remainder + digit (5.33 bytes a key) from the first level
    rem32 = ctx.stat("rem32_chunks") > 0          # second level writes / counting reads 4-byte remainders

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        wt = torch.tensor([windows], dtype=torch.int64, device=xdev)
        dist.all_reduce(wt, op=dist.ReduceOp.SUM)
        windows = int(wt.item())

    if rank == 0:
        value = windows / dt / 1e6
        klen = k + 1 if args.graph else k
        kbytes = 8 if 2 * klen <= 62 else 16
        b_per_window = algorithmic_bytes_per_window(klen, L, kbytes, 2 if args.graph else 1)
        # dominant kernel class by device time over the timed region
        # algorithmic bytes of one launch, from section 8(d)'s per-window figure
        # L/(L-k+1)*3/8 + 2*W*s: a partition / counting kernel moves the key term (one read and
        # one write of every key = 2*W per key); the extraction kernel's share is the read
        # term plus ONE write of the window's keys (W*s), counted per valid window
        def against_roofline(name):
            d = tim[name]
            launches = max(1, d["launches"])
            avg = d["ms"] / launches
            if name == "extract":
                keys_per_window = 2 if args.graph else 1
                per_unit = L / (L - klen + 1) * 3.0 / 8.0 + kbytes * keys_per_window
                units = w_local / args.steps / max(1, d["launches"] / args.steps)
            elif rem32:
                # the 32-bit-remainder form: the counting kernel reads a 4-byte remainder; the second level reads what
                # the first level wrote for it -- remainder + digit, twelve keys to a 64-byte granule = 5.33 bytes, in
                # the narrow form that has been the default since round 5, an 8-byte key before -- and writes the remainder
                per_unit = 4.0 if name == "reduce" else ((16.0 / 3.0 if narrow else 8.0) + 4.0)
                units = d["units"] / launches
            else:
                per_unit = (1.0 if name == "reduce" else 2.0) * kbytes
                units = d["units"] / launches
            gbs = per_unit * units / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
            return {"d": d, "avg_ms": avg, "per_unit": per_unit, "units": units, "achieved": gbs}

        # dominant kernel class by device time over the timed region; classes within 5 % of it count as
        # equally dominant (extraction and second level trade places from run to run) and the one that is
        # FURTHER from its roofline is the one reported
        longest = max(tim[n]["ms"] for n in tim if n != "emit")
        tied = [n for n in tim if n != "emit" and tim[n]["ms"] > 0 and tim[n]["ms"] >= 0.95 * longest]
        dom = min(tied, key=lambda n: against_roofline(n)["achieved"])
        r = against_roofline(dom)
        d, avg_ms, per_unit, per_launch_units, achieved = r["d"], r["avg_ms"], r["per_unit"], r["units"], r["achieved"]
        dev_ms = sum(v["ms"] for n, v in tim.items())
        tr, traffic_warning = measured_traffic(dom)
        traffic = tr["bytes_per_unit"] * per_launch_units if tr else None
        out = {
            "metric": ("M rho-mer windows/s (both strands counted) at k=%d, %d bp reads" if args.graph
                       else "M k-mers/s (canonical, counted) at k=%d, %d bp reads") % (k, L),
            "value": value,
            "unit": "M k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s: k=%d, %d x %d bp synthetic reads per GPU (genome %d bp, seed %d), in-HBM "
                                   "sort-count, %s emitted" % (config_name, k, nreads, L, genome_len, args.seed,
                                                               "Graph (edge SparseArray + counts)" if args.graph else "KmerSet SparseArray"),
                       "reads_per_gpu": nreads, "read_len": L, "k": k, "distinct_kmers": distinct,
                       "parallelism": "1 GPU" if world == 1 else "range-partition over %d ranks on %d GPU(s), %s all-to-all(v)"
                                      % (world, min(world, ndev), "RCCL" if args.backend == "nccl" else "gloo (host-staged)"),
                       "exchange": (args.exchange if use_dist else None), "route_parts": (args.route_parts or None)},
            "roofline": {"bound": "hbm", "kernel": {"extract": "extract1_part_kernel" if fused else "extract1_kernel",
                                                    "order": "canonical_map_kernel + radix passes over (key,count) pairs",
                                                    "hist": "radix_hist_kernel",
                                                    "scan": "scan_*_kernel", "scatter": "subpart32_kernel" if rem32 else "radix_onesweep_kernel",
                                                    "reduce": "seg_hash_reduce32b_kernel" if rem32 else "seg_hash_reduce_kernel"}.get(dom, dom),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": tr["source"] if tr else None,
                         "traffic_warning": traffic_warning,
                         "launch_avg_ms": avg_ms, "launches": d["launches"], "units_per_launch": per_launch_units,
                         "algorithmic_bytes_per_unit": per_unit,
                         "pipeline": {"algorithmic_bytes_per_kmer": b_per_window,
                                      "achieved": b_per_window * (windows / world) / (dt) / 1e9,
                                      "frac": b_per_window * (windows / world) / dt / 1e9 / HBM_PEAK_GBS},
                         "device_ms_per_step": {n: v["ms"] / args.steps for n, v in tim.items()},
                         # (the multi-GPU path: what the assembling rank's goss_gpu_emit_assemble took in the last step, host clock)
                         "assemble_ms": (ctx.stat("assemble_us") / 1e3) if use_dist else None,
                         "device_ms_total_per_step": dev_ms / args.steps},
        }
        # the other kernel classes against their own algorithmic bytes (same definitions)
        others = {}
        for name in ("extract", "scatter", "reduce"):
            if name == dom or tim[name]["ms"] <= 0:
                continue
            o = against_roofline(name)
            others[name] = {"ms_per_step": tim[name]["ms"] / args.steps, "launches_per_step": tim[name]["launches"] / args.steps,
                            "algorithmic_bytes_per_unit_per_launch": o["per_unit"], "achieved": o["achieved"],
                            "frac": o["achieved"] / HBM_PEAK_GBS}
        out["roofline"]["other_kernels"] = others
        # Beside the headline, never `value`: the same workload with the reads resident in HBM in the PACKED form (3 bits per
        # base: what the `goss` parser's threads hand over and SURVEY.md section 8(d)'s read term counts) -- packed on the
        # device by goss_gpu_pack_bases_device outside the timed region, counted by goss_gpu_push_packed_device: the same
        # kernels fetching 0.45 instead of 1.2 bytes per window and encoding nothing.
        if world == 1 and not use_dist and not args.graph and not args.no_packed:
            try:
                groups = (nbytes + 15) // 16
                dcodes = torch.empty(groups, dtype=torch.int32, device=device)
                dflags = torch.empty(groups, dtype=torch.int16, device=device)
                ctx.pack_bases_device(bases.data_ptr(), nbytes, dcodes.data_ptr(), dflags.data_ptr())

                def packed_step():
                    ctx.reset()
                    ctx.push_packed_device(dcodes.data_ptr(), dflags.data_ptr(), nbytes)
                    c = ctx.finish()
                    ctx.emit_device()
                    return c.windows, c.distinct

                for _ in range(max(1, args.warmup)):
                    packed_step()
                ctx.timing(reset=True)
                barrier()
                tp = time.perf_counter()
                pw = pd = 0
                for _ in range(args.steps):
                    w, pd = packed_step()
                    pw += w
                barrier()
                pdt = time.perf_counter() - tp
                ptim = ctx.timing().as_dict()
                e_ms = ptim["extract"]["ms"] / max(1, ptim["extract"]["launches"])
                e_bytes = (L / (L - klen + 1) * 3.0 / 8.0 + kbytes) * (pw / args.steps) / max(1.0, ptim["extract"]["launches"] / args.steps)
                out["packed_input"] = {"value": pw / pdt / 1e6, "unit": out["unit"], "ms_per_step": pdt / args.steps * 1e3,
                                       "distinct_kmers": pd, "same_result_as_headline": bool(pd == distinct and pw == windows),
                                       "device_ms_per_step": {n: v["ms"] / args.steps for n, v in ptim.items()},
                                       "first_level": {"launch_avg_ms": e_ms, "achieved": e_bytes / (e_ms * 1e-3) / 1e9 if e_ms > 0 else 0.0,
                                                       "frac": e_bytes / (e_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if e_ms > 0 else 0.0},
                                       "pipeline_frac": b_per_window * pw / pdt / 1e9 / HBM_PEAK_GBS,
                                       "packed_fused_chunks": ctx.stat("packed_fused_chunks"),
                                       "note": "reads resident in HBM as 2-bit codes + non-base flags (packed by goss_gpu_pack_bases_device "
                                               "before the timed region); `value` above is the byte form, which encodes inside the timed region"}
                del dcodes, dflags
            except Exception as e:          # the headline line must not be lost to the side record
                out["packed_input"] = {"failed": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            note("headline done: %.1f ms per step; CPU baseline (child process, <= 420 s)" % (dt / args.steps * 1e3))
            out["cpu_baseline"] = cpu_baseline_child(k, L, genome_len, args.seed, args.cpu_sample_reads)
            # BASELINE.md holds no published figure for this metric: the ratio is against the CPU port timed on
            # this box's host cores just above (same synthetic workload, bounded sample)
            if "value" in out["cpu_baseline"]:
                out["vs_baseline"] = value / out["cpu_baseline"]["value"]
                out["vs_baseline_of"] = "cpu_baseline.value (oracle port, %d threads, this box)" % out["cpu_baseline"]["cores"]
        # beside the headline, never part of `value`: the end-to-end CLI on a bounded FASTQ sample, and
        # BASELINE config C4 (build-graph k = 55, 200 M x 150 bp reads) through the same library
        if world == 1 and not use_dist and not args.graph:
            ctx.close()
            if args.e2e_reads > 0:
                n_e2e = min(args.e2e_reads, nreads)
                del bases
                torch.cuda.empty_cache()
                bases = None
                note("end-to-end CLI record (children, each with a time limit)")
                out["e2e"] = e2e_record(n_e2e, L, genome_len, args.seed, k, max(1, min(os.cpu_count() or 1, 64)))
                # SURVEY.md section 8(d)'s own metric (first input byte -> last file closed), copied where a reader of
                # the roofline block sees it: never `value`
                e = out["e2e"]
                if "seconds" in e:
                    out["roofline"]["e2e"] = {"seconds": e["seconds"], "reads": e["reads"], "runs_seconds": e["runs_seconds"],
                                              "first_run_seconds": e.get("first_run_seconds"), "phases": e.get("phases"),
                                              "first_run_phases": e.get("first_run_phases"),
                                              "parse_only_seconds": e["parse_only_seconds"], "M_kmers_per_s": e["value"],
                                              "what": "goss build-kmer-set on the bench's reads as a FASTQ file, process start -> files closed"}
            if not args.no_extra and not (args.reads or args.genome):
                bases = None
                torch.cuda.empty_cache()
                note("C4 record")
                out["extra"] = c4_record(g, torch, device, dev_index)
                torch.cuda.empty_cache()
                note("multi-GPU probe (two children, <= 300 s each)")
                out["multi_gpu_probe"] = multi_gpu_probe(8)
                note("side records done")
        # RCCL writes a version banner through C stdio; flush it so that the JSON line is last
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)

    ctx.close()          # (idempotent)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
